"""The N-rank routes, rehearsed with one rank on the 1-GPU box: (1) the native driver apps/batch_frames_mgpu (one process,
one vo_ctx + one host thread per device, ncclCommInitAll, one ncclAllGather of the poses per pass); (2) the route the
scaling run takes through bench.py: `python bench.py --gpus N` starting its own ranks (torch.distributed.run, RCCL
process group, all-gather of the poses).  N > 1 cannot run on this box -- the same code paths, one rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "apps", "bin")


def test_native_multi_gpu_driver_with_one_rank():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # 24 pairs x 3000 points, 20 rounds, 2 timed passes, in calls of 10 frames (24 = 10 + 10 + 4: the per-call bookkeeping)
    r = subprocess.run([os.path.join(BIN, "batch_frames_mgpu"), "1", "24", "3000", "20", "2", "10"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["pairs_total"] == 24 and d["bad_frames"] == 0 and d["gather_mismatches"] == 0
    assert d["worst_pose_err"] < 2e-3 and d["frames_per_sec"] > 0
    # more GPUs than the node has: a clean refusal, no hang
    r = subprocess.run([os.path.join(BIN, "batch_frames_mgpu"), "64", "64", "1000"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2 and "GPUs asked for" in r.stderr


def test_native_sequence_driver_with_one_rank(tmp_path):
    """apps/sequence_mgpu (second row: the consecutive pairs of a real sequence matched in blocks on the node's
    GPUs, counts + padded pair lists all-gathered with RCCL, the chain on rank 0) on the reference's dataset with ONE rank:
    every file it writes must equal, byte for byte, what `vo_complete --resident --match-up-front` writes -- fast and
    reference-order arithmetic."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    data = os.path.join(ROOT, "tests", "golden", "example_data", "data")
    for flags in ([], ["--exact"]):
        a, b = tmp_path / ("a" + "".join(flags)), tmp_path / ("b" + "".join(flags))
        a.mkdir(); b.mkdir()
        r = subprocess.run([os.path.join(BIN, "sequence_mgpu"), data, str(a), "1", "100"] + flags, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == 1 and d["frames"] == 121 and d["pairs_per_rank"] == 120 and d["matches_total"] == d["matches_found_by_the_ranks"] > 5000
        r2 = subprocess.run([os.path.join(BIN, "vo_complete"), data, str(b), "100", "--resident", "--match-up-front"] + flags,
                            capture_output=True, text=True, timeout=600, env=env)
        assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr
        for f in ("poses_raw.txt", "trajectory_est_complete.txt", "trajectory_est_data.txt", "map.txt", "map_appearances.txt"):
            assert (a / f).read_bytes() == (b / f).read_bytes(), (f, flags)
        assert len((a / "poses_raw.txt").read_text().splitlines()) == 121
    r = subprocess.run([os.path.join(BIN, "sequence_mgpu"), data, str(tmp_path), "64"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2 and "GPUs asked for" in r.stderr


def test_native_drivers_with_three_ranks_rehearsed_on_the_one_gpu(tmp_path):
    """VO_MGPU_SHARE_GPU=1: the native drivers with THREE ranks -- each its own context and host thread, all on this box's one
    GPU, the all-gathers staged through the host (RCCL refuses several ranks on one device).  Uneven blocks (13 pairs = 5 + 4 + 4;
    120 consecutive pairs = 40 + 40 + 40), padding rows, per-call slicing, own-block / foreign-block checks, the row mapping of the
    gathered pair lists and the chain on them run as they will on a multi-GPU node; the sequence driver's files must still equal
    `vo_complete --resident --match-up-front` byte for byte."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VO_MGPU_SHARE_GPU="1")
    r = subprocess.run([os.path.join(BIN, "batch_frames_mgpu"), "3", "13", "3000", "20", "2", "2"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 3 and d["pairs_total"] == 13 and d["bad_frames"] == 0 and d["gather_mismatches"] == 0 and "rehearsal" in d
    assert d["worst_pose_err"] < 2e-3
    data = os.path.join(ROOT, "tests", "golden", "example_data", "data")
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    r = subprocess.run([os.path.join(BIN, "sequence_mgpu"), data, str(a), "3", "100"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 3 and d["pairs_per_rank"] == 40 and d["matches_total"] == d["matches_found_by_the_ranks"] > 5000 and "rehearsal" in d
    env1 = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r2 = subprocess.run([os.path.join(BIN, "vo_complete"), data, str(b), "100", "--resident", "--match-up-front"], capture_output=True,
                        text=True, timeout=600, env=env1)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr
    for f in ("poses_raw.txt", "trajectory_est_complete.txt", "map.txt", "map_appearances.txt"):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f
    # seven ranks, 120 pairs = 18 x 1 + 17 x 6: blocks of different sizes, padding rows in the gathered lists
    c = tmp_path / "c"; c.mkdir()
    r = subprocess.run([os.path.join(BIN, "sequence_mgpu"), data, str(c), "7", "100"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    assert (c / "poses_raw.txt").read_bytes() == (b / "poses_raw.txt").read_bytes()


def test_bench_self_launch_route_with_one_rank():
    env = dict(os.environ, VO_BENCH_FORCE_LAUNCH="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-extras", "--steps", "20", "--warmup", "3"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["value"] > 10000 and d["pose_err_vs_gt"] < 1e-3
    # and the sharded frame legs under the same launcher (weak: 8 pairs per rank; strong: 12 pairs over the ranks)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "1", "--points", "4000",
                        "--legs", "frame", "--frame-steps", "2", "--strong-pairs", "12", "--strong-per-call", "5", "--gen-workers", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ranks_seen"] == 1 and d["batched_frames"]["frames_total"] == 200
    assert d["batched_frames_strong"]["pairs_total"] == 12 and d["batched_frames_strong"]["calls_per_pass"] == 3


def test_two_ranks_rehearsed_on_the_one_gpu():
    """`python bench.py --gpus 2` end to end with TWO ranks -- both on this box's one GPU (VO_BENCH_SHARE_GPU=1: gloo collectives
    staged through the host, because RCCL refuses two ranks on one device): the self-launch, the per-rank sharding (13 pairs =
    7 + 6: padded blocks), both gathers and every slice check run exactly as they will on a multi-GPU node.  The rates are not
    scaling numbers and the line says so."""
    env = dict(os.environ, VO_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--points", "4000",
                        "--legs", "frame", "--frame-steps", "2", "--strong-pairs", "13", "--strong-per-call", "4", "--gen-workers", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and "rehearsal" in d
    assert d["value"] > 10000 and d["pose_err_vs_gt"] < 1e-3
    assert d["batched_frames"]["frames_total"] == 400 and d["batched_frames"]["n_gpus"] == 2
    s = d["batched_frames_strong"]
    assert s["pairs_total"] == 13 and s["pairs_this_rank"] == 7 and s["calls_per_pass"] == 2 and s["n_gpus"] == 2
