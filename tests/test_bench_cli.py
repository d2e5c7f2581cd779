"""bench.py host logic that needs no GPU: `--gpus N` starts its own ranks (or fails cleanly when the node has
fewer GPUs), the frame-level algorithmic bytes are SURVEY 8(d)'s, the strong-scaling partition is contiguous."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_n_without_launcher_fails_cleanly_when_the_node_has_fewer_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: the launch would really run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "needs 2 GPUs" in r.stderr and "torch.distributed.run" not in r.stderr
    assert r.stdout.strip() == ""                      # no JSON line, no traceback


def test_frame_bytes_and_partition(vo):
    b = _bench()
    assert abs(b._frame_alg_bytes(50000, 50) - 65.4e6) < 1e3          # VERDICT / SURVEY 8(d): 65.4 MB per frame
    from importlib import import_module
    vdist = import_module("visual_odometry_amd.dist")
    for world in (1, 2, 4, 8):
        blocks = [vdist.shard_range(1600, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == 1600
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
        assert all(hi - lo == 1600 // world for lo, hi in blocks)
    g = b._PairGen(300, 7)
    fps = g(0, 2)
    assert len(fps) == 2 and len(fps[0]["ref_app"]) == 300
    ref = vo.synth.frame_pair(300, seed=4000 + 8)
    assert (fps[1]["ref_app"] == ref["ref_app"]).all() and (fps[1]["X_gt"] == ref["X_gt"]).all()


def test_result_line_is_json_for_every_leg_selection():
    """cpu_leg and exact_leg carry an ndarray under "_pose"; whichever legs ran, the assembled dict must
    serialise (with `--legs cpu` the array used to stay in the dict and json.dumps raised after all the GPU work)."""
    import json
    import numpy as np
    b = _bench()
    pose = np.eye(4, dtype=np.float32)
    shapes = {
        "cpu only": {"cpu_baseline": {"value": 860.0, "_pose": pose.copy()}},
        "exact only": {"exact_mode": {"iters_per_sec": 5500.0, "_pose": pose.copy()}},
        "both": {"exact_mode": {"iters_per_sec": 5500.0, "_pose": pose.copy()}, "cpu_baseline": {"value": 860.0, "_pose": pose.copy()}},
        "both, poses differ": {"exact_mode": {"iters_per_sec": 1.0, "_pose": pose.copy()}, "cpu_baseline": {"value": 2.0, "_pose": pose + 1}},
        "neither": {"metric": "x"},
    }
    for name, out in shapes.items():
        b._relate_exact_and_cpu(out)
        line = json.dumps(out)                                  # must not raise
        assert "_pose" not in line, name
    assert shapes["both"]["exact_mode"]["bit_identical_to_cpu_baseline"] is True
    assert abs(shapes["both"]["exact_mode"]["vs_cpu_baseline"] - 5500.0 / 860.0) < 1e-9
    assert shapes["both, poses differ"]["exact_mode"]["bit_identical_to_cpu_baseline"] is False
    assert "bit_identical_to_cpu_baseline" not in shapes["exact only"]["exact_mode"]


def test_strong_pairs_fewer_than_ranks_is_rejected_up_front():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--strong-pairs", "3"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "every rank needs at least one pair" in r.stderr and r.stdout.strip() == ""


def test_committed_counter_summaries_feed_the_roofline_objects():
    """bench.py takes counter-derived figures (HBM traffic, executed VALU instructions) from the newest summaries under
    profiles/: the kernel names and launch geometries it asks for must exist there, or a roofline object silently loses its
    `traffic` / `achieved`"""
    b = _bench()
    traffic, note = b._pmc_traffic("vo::picp_round_kernel<true, false, true, false>", 256 * ((50000 + 255) // 256))
    assert traffic is not None and 0.3e6 < traffic < 3e6, note            # one round over a 50k pair: ~0.75 MB
    t2, note2 = b._pmc_traffic("vo::picp_batch_kernel<true, false>", 768 * 200)
    assert t2 is not None and 5e9 < t2 < 12e9, note2                      # 200 x 50k x 50 rounds: ~9 GB
    call, _ = b._pmc_frames_call()
    assert call is not None and 10e9 < call < 20e9                        # one 200-frame call: ~13.9 GB
    for kernels, pick in ((("vo::match_init_kernel", "vo::match_kernel<false>", "vo::match_count_kernel", "vo::match_scatter_kernel"), min),
                          (("vo::match_minmax_kernel", "vo::match_bucket_hist_kernel", "vo::match_bucket_offsets_kernel",
                            "vo::match_bucket_place_kernel", "vo::match_pruned_kernel", "vo::match_count_kernel", "vo::match_scatter_kernel"), min),
                          (b.MATCHER_CHAIN, max)):       # the batched call's matcher stage: exact-duplicate pass (+ the skipped search)
        insts, util, why = b._pmc_valu(kernels, pick)
        assert insts is not None and insts > 1e6 and 0.5 < util <= 1.0, (kernels, why)
    m = b._matcher_roofline(200, 50000, 0.5e-3)
    assert m["bound"] == "hbm" and m["traffic"] is not None and 0.5 < m["traffic_over_algorithmic"] < 3.0, m
    r = b._valu_roofline(("vo::match_init_kernel", "vo::match_kernel<false>"), 0.6e-3, "test", min)
    assert r["bound"] == "valu" and 0.3 < r["frac"] < 0.8                 # the full scan: about half the FP32 vector peak
