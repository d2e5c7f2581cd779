"""configs 3/5: the C++ drivers `whole_test` and `vo_complete` + `evaluate` on the
GPU path, against the README metrics and the oracle-side run of the same loop."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import vo_pipeline as vp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "apps", "bin")
DATA = os.path.join(ROOT, "tests", "golden", "example_data", "data")
README = dict(inv_ratio=0.47337, rmse_points=0.184143, rmse_pos=0.145332)


def test_whole_test_app():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    for seed in ("3", "5", "11", "12", "13", "14", "15", "16"):     # general motions: rotation up to 0.2 rad, translation 0.5
        r = subprocess.run([os.path.join(BIN, "whole_test"), seed, "4000"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr                # initialisation within 1e-4 of the generating motion, PICP 1e-2
        assert "EPIPOLAR" in r.stdout and "PICP" in r.stdout


def _vo_complete(out_dir, *flags):
    os.makedirs(out_dir, exist_ok=True)
    r = subprocess.run([os.path.join(BIN, "vo_complete"), DATA, str(out_dir), *flags], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    poses = np.loadtxt(os.path.join(out_dir, "poses_raw.txt"), dtype=np.float64).astype(np.float32).reshape(-1, 4, 4)
    counts = np.array(re.findall(r"^meas-\d+\.dat: (\d+) matches, (\d+) model correspondences, (\d+) inliers", r.stdout, flags=re.M), dtype=int)
    return poses, counts


def test_vo_complete_on_example_data(tmp_path, o32):
    """BASELINE configs[4] in the default (fast) arithmetic.  The parity evidence for this dataset is the reference-order
    run (tests/test_gpu_exact.py: every count and every pose of the chain bit for bit).  The fast mode is judged against
    THAT run frame by frame while the chain is young: the tail of this sequence is chaotic -- 119 chained solves with 6-32
    inliers each, where a last-bit change of one pose flips a borderline z_far gate some frames later and moves the end-to-end
    RMSE figures by tens of percent within the float32 oracle itself -- so end-to-end bounds on the fast mode's RMSE say
    nothing (round-2 review, item 7)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    fast, c_fast = _vo_complete(tmp_path / "fast")
    exact, c_exact = _vo_complete(tmp_path / "exact", "--exact")
    assert fast.shape == exact.shape == (121, 4, 4) and c_fast.shape == c_exact.shape == (119, 3)
    # appearance matches do not depend on the pose: equal on all 119 frames; the joined correspondences follow the previous
    # frame's triangulation: equal while the chain is young
    assert np.array_equal(c_fast[:, 0], c_exact[:, 0])
    assert np.array_equal(c_fast[:40, 1], c_exact[:40, 1])
    # inlier counts of the last round: the default mode's fused arithmetic leaves every projection within an ulp or two of
    # the reference-order value, so a correspondence that lies within that rounding of a gate or of the chi^2 threshold may
    # fall on the other side (measured: one correspondence in frame 12 of the first 45 frames) -- never more than one per
    # frame, and rarely
    dn = np.abs(c_fast[:40, 2] - c_exact[:40, 2])
    assert dn.max() <= 1 and (dn != 0).sum() <= 3, (c_fast[:40, 2], c_exact[:40, 2])
    # per-frame relative pose against the reference-order run, first 40 frames (as tests/test_gpu_fullsize.py does for
    # config 3): rounding-level
    d = np.abs(fast[:41] - exact[:41]).reshape(41, -1).max(1)
    assert d.max() < 5e-4, d
    # the evaluation's scale (a median over all frames, hence robust) still reproduces the README's 1/r_t
    e = subprocess.run([os.path.join(BIN, "evaluate"), DATA, str(tmp_path / "fast")], capture_output=True, text=True, timeout=60)
    assert e.returncode == 0, e.stdout + e.stderr
    val = {k: float(v) for k, v in re.findall(r"^(.*?):\s*([-0-9.e+]+)", e.stdout, flags=re.M)}
    assert abs(val["ratio used for map correction"] - README["inv_ratio"]) < 0.015 * README["inv_ratio"]
    # and the oracle-side run of the loop agrees with the fast mode's first dozen robot positions
    res = vp.run_vo_complete(DATA, rounds=100, o=o32)
    est = np.loadtxt(os.path.join(tmp_path / "fast", "trajectory_est_complete.txt"))
    ref = np.array([T[:3, 3] for T in vp.robot_trajectory(res["trajectory"], res["H"])])
    assert est.shape == ref.shape == (121, 3)
    assert np.abs(est[:12] - ref[:12]).max() < 2e-3
    assert np.array_equal(c_fast[:, 0], np.array(res["stats"], dtype=int)[:, 0])


def test_cpp_batch_frames_app():
    """apps/batch_frames.cpp: config 4 driven from plain C++ over the C ABI (vo_frames_batch_dev); exits 0 only when
    every frame found all its matches / joins / inliers and its pose is the generator's ground truth."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    r = subprocess.run([os.path.join(BIN, "batch_frames"), "12", "3000", "20", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "12 frames x 3000 points" in r.stdout and "missing match/join/inlier: 0" in r.stdout
