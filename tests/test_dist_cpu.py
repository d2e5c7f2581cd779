"""N>1 host logic on CPU: two gloo ranks shard independent frame pairs, run a
rank-dependent stand-in for the per-pair result and gather the poses exactly as
bench.py does over RCCL.  No GPU, no oracle: only the sharding/gather plumbing."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    import __graft_entry__ as g
    g.load_package()
    from importlib import import_module
    vdist = import_module("visual_odometry_amd.dist")
    rank, local_rank, world = vdist.env_rank_world()
    dist = vdist.init("gloo")
    n_pairs = 7                                   # not divisible: blocks of 4 and 3
    lo, hi = vdist.shard_range(n_pairs, rank, world)
    assert (lo, hi) == ((0, 4), (4, 7))[rank]
    # equal-sized blocks for the all-gather: pad to the largest block
    blk = -(-n_pairs // world)
    local = torch.zeros((blk, 16), dtype=torch.float32)
    for i, p in enumerate(range(lo, hi)):
        local[i] = torch.arange(16, dtype=torch.float32) + 100.0 * p      # "pose" of pair p
    allp = vdist.gather_poses(local)
    assert allp.shape == (world * blk, 16)
    for r in range(world):
        l2, h2 = vdist.shard_range(n_pairs, r, world)
        for i, p in enumerate(range(l2, h2)):
            assert torch.equal(allp[r * blk + i], torch.arange(16, dtype=torch.float32) + 100.0 * p)
    # ragged gather (up-front matching of a sequence): item p carries p+1 pairs, blocks of 4 and 3 items
    mine = [torch.stack([torch.arange(p + 1, dtype=torch.int32), torch.full((p + 1,), p, dtype=torch.int32)], 1)
            for p in range(lo, hi)]
    every = vdist.gather_ragged(mine)
    assert len(every) == n_pairs
    for p, it in enumerate(every):
        assert it.shape == (p + 1, 2) and int(it[:, 1].min()) == p == int(it[:, 1].max())
        assert torch.equal(it[:, 0], torch.arange(p + 1, dtype=torch.int32))
    # the strong-scaling leg's bookkeeping (bench.frame_throughput_strong): equal-sized padded blocks, own-slice check,
    # global order after the gather -- uneven (7 = 4 + 3), even (8), and one pair per rank (2)
    for P in (7, 8, 2):
        plan = vdist.StrongPlan(P, rank, world)
        assert (plan.lo, plan.hi) == vdist.shard_range(P, rank, world) and plan.blk == -(-P // world)
        buf = torch.full((plan.blk, 16), -1.0)                # padding rows keep -1
        for i, p in enumerate(range(plan.lo, plan.hi)):
            buf[i] = torch.arange(16, dtype=torch.float32) + 100.0 * p
        allp = vdist.gather_poses(buf)
        assert allp.shape == (world * plan.blk, 16)
        r0, r1 = plan.own_rows()
        assert torch.equal(allp[r0:r1], buf[:plan.n_local])
        g_all = plan.global_order(allp)
        assert g_all.shape == (P, 16)
        for p in range(P):
            assert torch.equal(g_all[p], torch.arange(16, dtype=torch.float32) + 100.0 * p)
    try:
        vdist.StrongPlan(1, rank, world)
        raise AssertionError("one pair over two ranks must be rejected")
    except ValueError:
        pass
    t = vdist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    assert t == float(world)
    # bench.py's per-rank rows (compute time, gather time, units): every rank's values on every rank, rank-major
    rows = vdist.gather_values([0.5 + rank, 2.0 * rank, 200.0], torch.device("cpu"))
    assert rows == [[0.5, 0.0, 200.0], [1.5, 2.0, 200.0]]
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_gloo_ranks_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {rank} ok" in o


def test_shard_range_partitions():
    import __graft_entry__ as g
    g.load_package()
    from importlib import import_module
    vdist = import_module("visual_odometry_amd.dist")
    for n in (0, 1, 7, 200, 1600):
        for w in (1, 2, 4, 8):
            seen = []
            for r in range(w):
                lo, hi = vdist.shard_range(n, r, w)
                assert 0 <= lo <= hi <= n and hi - lo in (n // w, n // w + 1)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert vdist.shard_range(1600, 3, 8) == (600, 800)      # BASELINE config 4: 200 pairs per rank
