"""The reference's known-association programs (src/tests/picp_real_data_allKnown.cpp, initialization_real_data.cpp,
vo_daKnown.cpp) on the GPU path, on the reference's own data directory: against the GROUND TRUTH the data ships
(trajectory.dat, world.dat) and against the oracle's run of the same loops -- bit for bit with the solver in
reference-order arithmetic.  Through the ctypes binding and through the C++ drivers in apps/."""
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


def _poses_raw(path):
    a = np.loadtxt(path, dtype=np.float64).reshape(-1, 4, 4)
    return a.astype(np.float32)


def _known_real_gpu(vo, ctx, rounds, exact):
    files, K, H, (rows, cols, zn, zf) = vp._dataset(DATA)
    world, _ = vp.read_world(os.path.join(DATA, "world.dat"))
    X = vp.iso_inv(H.astype(np.float64)).astype(np.float32)
    pts = world.copy()
    s = vo.PICPSolver(ctx)
    s.setKernelThreshold(10000.0)
    s.setExact(exact)
    traj, stats = [], []
    for f in files:
        meas, _, ids = vp.read_meas(os.path.join(DATA, f))
        pts = vo.transform_points(X, pts, ctx=ctx)
        corr = np.stack([np.arange(len(ids)), ids], axis=1).astype(np.int32)
        s.init(vo.Camera(rows, cols, zn, zf, K, np.eye(4), ctx=ctx), pts, meas)
        s.solve(corr, False, rounds)
        X = s.camera().worldInCameraPose().copy()
        traj.append(X)
        stats.append((len(corr), s.numInliers()))
    s.close()
    return traj, stats, H


@pytest.mark.parametrize("exact", [True, False])
def test_picp_known_real_binding(vo, ctx, o32, exact):
    traj, stats, H = _known_real_gpu(vo, ctx, 200, exact)
    err, _ = vp.gt_errors(DATA, traj, H)
    assert len(traj) == 121 and err.max() < 1e-4, err.max()                # the ground truth of trajectory.dat
    r = vp.run_picp_known_real(DATA, rounds=200, o=o32)
    assert stats == r["stats"]                                             # inlier counts, every frame
    if exact:
        assert np.array_equal(np.array(traj), np.array(r["trajectory"]))   # 121 chained solves x 200 rounds: same bits
    else:
        assert np.abs(np.array(traj) - np.array(r["trajectory"])).max() < 2e-5


def test_picp_known_real_app(tmp_path, o32):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    r = vp.run_picp_known_real(DATA, rounds=1000, o=o32)                   # the reference's round count
    for flags in ([], ["--exact"]):
        out = tmp_path / ("exact" if flags else "fast")
        out.mkdir()
        p = subprocess.run([os.path.join(BIN, "picp_known_real"), DATA, str(out), "1000"] + flags, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr                # the driver itself checks the ground truth (2e-4)
        dev = float(re.search(r"max abs deviation from the ground-truth trajectory: ([-0-9.e+]+)", p.stdout).group(1))
        assert dev < 1e-4
        poses = _poses_raw(out / "poses_raw.txt")
        assert poses.shape == (121, 4, 4)
        if flags:
            assert np.array_equal(poses, np.array(r["trajectory"]))
        else:
            assert np.abs(poses - np.array(r["trajectory"])).max() < 2e-5
        est = np.loadtxt(out / "trajectory_est.txt")
        ref = np.array([T[:3, 3] for T in vp.robot_trajectory(r["trajectory"], r["H"])])
        assert est.shape == (121, 3) and np.abs(est - ref).max() < 1e-4


def test_real_init_app_and_binding(vo, ctx, tmp_path, o32):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    p = subprocess.run([os.path.join(BIN, "real_init"), DATA, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr                           # rotation 1e-4, direction 1e-3, landmarks 5e-3 vs ground truth
    m = re.search(r"rotation error ([-0-9.e+]+), translation direction error ([-0-9.e+]+), median landmark error ([-0-9.e+]+)", p.stdout)
    assert float(m.group(1)) < 2e-5 and float(m.group(2)) < 5e-5 and float(m.group(3)) < 2e-3
    assert "115 correspondences, 115 triangulated" in p.stdout
    r = vp.run_real_init(DATA, o32)
    X = vo.estimate_transform(r["K"], r["corr"], r["p0"], r["p1"], ctx=ctx)
    assert np.abs(X - r["X"]).max() < 2e-5                                  # Jacobi SVD here, LAPACK in the oracle
    tri = np.loadtxt(tmp_path / "triangulated.txt")
    xyz, pairs, _ = o32.triangulate(r["K"], X, r["corr"], r["p0"], r["p1"])
    assert tri.shape == (115, 3) and np.allclose(tri, o32.transform_points(r["H"], xyz), rtol=0, atol=2e-5)   # text file: 6 digits


def test_vo_da_known_app(tmp_path, o32):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    out = tmp_path / "exact"; out.mkdir()
    p = subprocess.run([os.path.join(BIN, "vo_da_known"), DATA, str(out), "100", "--exact"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr
    poses = _poses_raw(out / "poses_raw.txt")
    assert poses.shape == (121, 4, 4)
    # the oracle's loop from the library's first relative pose (host double arithmetic on both sides, equal to 2e-5):
    # every count and every pose of the chain bit for bit
    r = vp.run_vo_da_known(DATA, rounds=100, o=o32, X0=poses[1])
    counts = np.array(re.findall(r"^meas-\d+\.dat: (\d+) associated, (\d+) model correspondences, (\d+) inliers", p.stdout, flags=re.M), dtype=int)
    assert np.array_equal(counts, np.array(r["stats"], dtype=int))
    assert np.array_equal(poses, np.array(r["trajectory"]))
    scale = float(re.search(r"median translation ratio, inverted: ([-0-9.e+]+)", p.stdout).group(1))
    assert abs(scale - 0.47337) < 3e-4                                      # README 1/r_t
    assert len(open(out / "time_known.txt").read().split()) == 119
    # the id association gives the pairs of the appearance matcher on this data: vo_complete --exact walks the same chain
    out2 = tmp_path / "complete"; out2.mkdir()
    q = subprocess.run([os.path.join(BIN, "vo_complete"), DATA, str(out2), "100", "--exact"], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0
    assert np.array_equal(_poses_raw(out2 / "poses_raw.txt"), poses)


def test_compute_corr_kdtree_and_read_data_apps():
    """The remaining programs of the reference's src/tests/ on the GPU path: compute_corr (appearance matcher = id association on
    every consecutive pair of the data directory), the kd-tree test (approximate against exact search) and read_data_test."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    p = subprocess.run([os.path.join(BIN, "compute_corr"), DATA], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr
    assert "120 consecutive frame pairs" in p.stdout and "equals the id association everywhere" in p.stdout
    for args in (["1"], ["2", "500", "200", "10"], ["3", "5000", "400", "20"], ["4", "7", "10", "10"]):
        q = subprocess.run([os.path.join(BIN, "kdtree_test")] + args, capture_output=True, text=True, timeout=120)
        assert q.returncode == 0, q.stdout[-1500:] + q.stderr
        assert "tree ok" in q.stdout and "FAST" in q.stdout
    for seed in ("3", "5", "11", "12", "13"):              # initialization_test.cpp: rotation 1e-4, translation ratios consistent
        t = subprocess.run([os.path.join(BIN, "init_test"), seed, "4000"], capture_output=True, text=True, timeout=60)
        assert t.returncode == 0, t.stdout + t.stderr
    r = subprocess.run([os.path.join(BIN, "read_data_test"), DATA], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "121 measurement files" in r.stdout and "world.dat: 1000 landmarks" in r.stdout, r.stdout
