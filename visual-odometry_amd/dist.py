"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).  The path shards by independent frame
pairs -- no data-path collective -- and ends with ONE gather of the SE(3) poses
(16 floats per pair: latency-bound, 64 B..12.8 KB per rank)."""
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
    out = torch.empty((world * local_poses.shape[0],) + tuple(local_poses.shape[1:]),
                      dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(out, local_poses.contiguous())
    return out


def max_over_ranks(value: float, device) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
