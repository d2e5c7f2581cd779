"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).  The path shards by independent frame
pairs -- no data-path collective -- and ends with ONE gather of the SE(3) poses
(16 floats per pair: latency-bound, 64 B..12.8 KB per rank).

A real sequence is serial in its pose chain (replicas only), but its matcher stage
depends on the appearances alone: all (t-1, t) pairs can be matched up front,
sharded over the ranks, and exchanged with `gather_ragged` (counts first, then one
padded all-gather) before one rank runs the chain (second row)."""
from __future__ import annotations

import os


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition: pair p -> rank floor(p / ceil(n/world)).
    Returns [lo, hi) for `rank`; blocks differ by at most one item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class StrongPlan:
    """Bookkeeping of the strong-scaling leg (BASELINE configs[3]): P pairs in total, rank r owns the contiguous block
    shard_range(P, r, world); the all-gather needs equal-sized blocks, so every rank's pose buffer has `blk` rows (the
    largest block) of which the first n_local are live.  P < world would leave a rank without a pair: rejected."""

    def __init__(self, n_pairs: int, rank: int, world: int):
        if n_pairs < world:
            raise ValueError(f"{n_pairs} pairs cannot be sharded over {world} ranks (every rank needs at least one)")
        self.P, self.rank, self.world = n_pairs, rank, world
        self.lo, self.hi = shard_range(n_pairs, rank, world)
        self.n_local = self.hi - self.lo
        self.blk = shard_range(n_pairs, 0, world)[1]          # rank 0 holds a largest block

    def own_rows(self):
        """rows of the gathered (world*blk, 16) tensor that hold THIS rank's live poses"""
        return self.rank * self.blk, self.rank * self.blk + self.n_local

    def global_order(self, gathered):
        """(world*blk, ...) gathered tensor -> (P, ...) in global pair order (the padding rows dropped)"""
        import torch
        parts = []
        for r in range(self.world):
            lo, hi = shard_range(self.P, r, self.world)
            parts.append(gathered[r * self.blk: r * self.blk + (hi - lo)])
        return torch.cat(parts, 0)


def init(backend: str, device_index: int | None = None):
    import torch.distributed as dist
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device_index is not None:
            import torch
            kw["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(backend=backend, **kw)
    return dist


def gather_poses(local_poses):
    """all-gather of equal-sized pose blocks: local (n_local,16) -> (world*n_local,16),
    rank-major, i.e. global pair order under shard_range with equal blocks."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if dist.get_backend() == "gloo" and local_poses.is_cuda:
        # rehearsal on a shared GPU (bench.py, VO_BENCH_SHARE_GPU=1): gloo gathers host tensors
        host = local_poses.detach().contiguous().cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host)
        return torch.cat(parts, 0).to(local_poses.device)
    out = torch.empty((world * local_poses.shape[0],) + tuple(local_poses.shape[1:]),
                      dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(out, local_poses.contiguous())
    return out


def max_over_ranks(value: float, device) -> float:
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_values(values, device):
    """every rank's list of floats on every rank: (world, len(values)) nested list, rank-major (bench.py's per-rank rows)"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [[float(x) for x in p.tolist()] for p in parts]


def gather_ragged(items, device=None):
    """All-gather of variable-length int32 pair lists: `items` is this rank's list of (n_i, 2) int32
    tensors (its contiguous block under shard_range); returns the list over ALL ranks in global item
    order.  Two collectives: the per-item counts (equal-sized blocks, padded with -1), then one padded
    all-gather of the pairs."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    device = device if device is not None else (items[0].device if items else torch.device("cpu"))
    n_local = torch.tensor([len(items)], dtype=torch.int64, device=device)
    n_all = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(n_all, n_local)
    blk = int(max(int(x.item()) for x in n_all))
    if blk == 0:
        return []
    counts = torch.full((blk,), -1, dtype=torch.int64, device=device)
    for i, it in enumerate(items):
        counts[i] = it.shape[0]
    all_counts = torch.empty((world * blk,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_counts, counts)
    width = max(int(all_counts.max().item()), 1)
    pad = torch.zeros((blk, width, 2), dtype=torch.int32, device=device)
    for i, it in enumerate(items):
        pad[i, : it.shape[0]] = it.to(device=device, dtype=torch.int32)
    allp = torch.empty((world * blk, width, 2), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(allp, pad)
    out = []
    for k in range(world * blk):
        c = int(all_counts[k].item())
        if c >= 0:
            out.append(allp[k, :c].clone())
    return out
