"""The per-element functors the kernels run (vo_math.h), compiled for the host
with g++, against the oracle.  Decision-making values must agree bit for bit;
accumulators (which use FMA) to rounding."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import rel_err
from oracle.oracle import Camera

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hc():
    so = os.path.join(HERE, "hostcheck", "libvo_hostcheck.so")
    src = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
    hdr = os.path.join(HERE, "..", "visual-odometry_amd", "csrc", "vo_math.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-Wno-unknown-pragmas", "-o", so, src])
    return C.CDLL(so)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def cm(M, n):
    return np.ascontiguousarray(np.asarray(M, np.float32).reshape(n, n).T).ravel()


def test_projection_is_bit_exact(hc, o32, vo):
    rng = np.random.default_rng(1)
    T = vo.synth.random_isometry(rng, 0.3, 0.5)
    cam = Camera(480, 640, 0, 10, vo.synth.K_REF, T)
    pts = vo.synth.random_points3d(rng, 4000) * np.float32([0.3, 0.3, 4.0])
    uv_o, _ = o32.project_points(cam, pts, keep_indices=True)
    uv = np.zeros(2, np.float32)
    n_ok = 0
    for i in range(len(pts)):
        ok = hc.hc_project_point(480, 640, 0, 10, p(cm(cam.K, 3)), p(cm(T, 4)), p(pts[i]), p(uv))
        if ok:
            n_ok += 1
            assert uv.tobytes() == uv_o[i].tobytes()
        else:
            assert uv_o[i, 0] == -1
    assert 100 < n_ok < len(pts)


def test_picp_term_and_update(hc, o32, o64, vo):
    fp = vo.synth.frame_pair(1500, seed=41, drop=0.05, distractors=5, model_drop=0.05)
    j = o32.join(o32.match(fp["ref_app"], fp["cur_app"]), fp["model_pairs"])
    cam = Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4))
    for thr, keep in ((10000.0, 0), (50.0, 0), (50.0, 1)):
        r32 = o32.picp_solve(cam, fp["model"], fp["cur_pts"], j, 1, thr, bool(keep))
        r64 = o64.picp_solve(cam, fp["model"], fp["cur_pts"], j, 1, thr, bool(keep))
        acc = np.zeros(30, np.float32)
        hc.hc_picp_accumulate(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], p(cm(fp["K"], 3)),
                              p(cm(np.eye(4), 4)), C.c_float(thr), keep, p(fp["model"]), p(fp["cur_pts"]),
                              p(j), len(j), p(acc))
        accp = np.zeros(30, np.float32)
        assert hc.hc_is_pinhole(p(cm(fp["K"], 3))) == 1
        hc.hc_picp_accumulate_pinhole(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], p(cm(fp["K"], 3)),
                                      p(cm(np.eye(4), 4)), C.c_float(thr), keep, p(fp["model"]), p(fp["cur_pts"]),
                                      p(j), len(j), p(accp))
        assert accp.tobytes() == acc.tobytes()          # pinhole specialisation == general 3x3 K, bit for bit
        # the batched solver's form (rejected terms zeroed through the 0 * x = 0 product instead of eight selects): same sums,
        # pinhole and general -- also with points whose rejected projection holds inf / NaN (z = 0, far outside the image)
        for general in (0, 1):
            accm = np.zeros(30, np.float32)
            hc.hc_picp_accumulate_mul0(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], p(cm(fp["K"], 3)), p(cm(np.eye(4), 4)),
                                       C.c_float(thr), keep, general, p(fp["model"]), p(fp["cur_pts"]), p(j), len(j), p(accm))
            assert np.array_equal(accm, acc), (thr, keep, general)
        bad = fp["model"].copy()
        bad[j[:40, 1], 2] = 0.0; bad[j[40:60, 1], 0] = 1e30; bad[j[60:70, 1], 2] = np.inf; bad[j[70:80, 1], 2] = 1e-40
        accb = np.zeros(30, np.float32); accn = np.zeros(30, np.float32)
        hc.hc_picp_accumulate(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], p(cm(fp["K"], 3)), p(cm(np.eye(4), 4)),
                              C.c_float(thr), keep, p(bad), p(fp["cur_pts"]), p(j), len(j), p(accb))
        hc.hc_picp_accumulate_mul0(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], p(cm(fp["K"], 3)), p(cm(np.eye(4), 4)),
                                   C.c_float(thr), keep, 0, p(bad), p(fp["cur_pts"]), p(j), len(j), p(accn))
        assert np.isfinite(accb).all() and np.array_equal(accn, accb)
        H = np.zeros((6, 6), np.float32)
        H[np.triu_indices(6)] = acc[:21]
        H = H + np.triu(H, 1).T
        # same sequential order as the oracle; only FMA rounding differs
        assert rel_err(H, r32["H"][0]) < 2e-6 and rel_err(acc[21:27], r32["b"][0]) < 2e-5
        assert rel_err(H, r64["H"][0]) < 3e-5
        assert acc[29] == r32["stats"][0, 2]                                   # inlier count: exact
        assert abs(acc[27] - r32["stats"][0, 0]) <= 1e-6 * max(1, r32["stats"][0, 0])
        assert abs(acc[28] - r32["stats"][0, 1]) <= 1e-6 * max(1, r32["stats"][0, 1])
        T1 = np.zeros(16, np.float32); Hd = np.zeros(36, np.float32); b = np.zeros(6, np.float32)
        hc.hc_picp_update(p(acc), C.c_float(1.0), p(cm(np.eye(4), 4)), p(T1), p(Hd), p(b))
        assert np.abs(T1.reshape(4, 4).T - r32["T_trace"][0]).max() < 2e-5
        assert np.array_equal(Hd.reshape(6, 6).T, H + np.eye(6, dtype=np.float32))


def test_ldlt6_bit_exact_with_oracle(hc, o32):
    rng = np.random.default_rng(2)
    for k in range(50):
        A = rng.normal(size=(9, 6)).astype(np.float32)
        S = (A.T @ A).astype(np.float32)
        if k % 3 == 0:
            S = S * np.float32(10.0) ** rng.integers(-3, 6, 6)[None, :] * np.float32(10.0) ** rng.integers(-3, 6, 6)[:, None]
            S = ((S + S.T) / 2).astype(np.float32)
        S += np.eye(6, dtype=np.float32)
        b = rng.normal(size=6).astype(np.float32)
        x = np.zeros(6, np.float32)
        hc.hc_ldlt6(p(np.ascontiguousarray(S.T).ravel()), p(b), p(x))
        assert x.tobytes() == o32.ldlt_solve(S, b).tobytes()
        # the kernels' variant (pivot order derived up front, FMA updates, one
        # reciprocal per column): same algorithm, rounding-level differences only
        xp = np.zeros(6, np.float32)
        hc.hc_ldlt6_perm(p(np.ascontiguousarray(S.T).ravel()), p(b), p(xp))
        S64, b64 = S.astype(np.float64), b.astype(np.float64)
        # backward error of both variants at float32 level (conditioning-independent)
        for sol in (x, xp):
            res = np.abs(S64 @ sol - b64).max()
            assert res <= 2e-5 * (np.abs(S64).sum(1).max() * np.abs(sol).max() + np.abs(b64).max())
        if k % 3 != 0:   # well conditioned: forward errors comparable
            x64 = np.linalg.solve(S64, b64)
            assert np.abs(xp - x64).max() <= 8 * max(np.abs(x - x64).max(), 1e-6 * np.abs(x64).max())
    z = np.zeros(6, np.float32)
    hc.hc_ldlt6(p(np.zeros(36, np.float32)), p(np.ones(6, np.float32)), p(z))
    assert np.all(z == 0)
    hc.hc_ldlt6_perm(p(np.zeros(36, np.float32)), p(np.ones(6, np.float32)), p(z))
    assert np.all(z == 0)
    # ties on the diagonal (H = I + 0): identity permutation, exact solution
    I6 = np.eye(6, dtype=np.float32) * np.float32(4.0)
    hc.hc_ldlt6_perm(p(I6.ravel()), p(np.arange(6, dtype=np.float32)), p(z))
    assert np.array_equal(z, np.arange(6, dtype=np.float32) / np.float32(4.0))


def test_ldlt6_without_pivoting_on_positive_definite_systems(hc, o32):
    """The round kernels' solve since round 3 eliminates in the natural order: H = sum(lambda J^T J) + I is positive
    definite, where LDL^T is backward stable in any order.  On SPD systems: float32-level backward error like the pivoted
    variants; bit-identical to the pivot-simulating variant whenever the diagonal already descends (its permutation is the
    identity then); on a PICP-shaped H (rotation block 1e3 x the translation block) the forward error stays comparable."""
    rng = np.random.default_rng(5)
    for k in range(200):
        A = rng.normal(size=(9, 6)).astype(np.float32)
        if k % 2:                                            # PICP-like scaling: columns 3..5 (rotation) dominate
            A[:, 3:] *= np.float32(10.0) ** rng.uniform(1, 2.5)
        S = (A.T @ A).astype(np.float32) + np.eye(6, dtype=np.float32)
        if k % 5 == 0:                                       # descending diagonal: Eigen's order is the natural one
            order = np.argsort(-np.diag(S), kind="stable")
            S = np.ascontiguousarray(S[np.ix_(order, order)])
        b = rng.normal(size=6).astype(np.float32)
        xo, xp, xe = np.zeros(6, np.float32), np.zeros(6, np.float32), np.zeros(6, np.float32)
        hc.hc_ldlt6_ordered(p(np.ascontiguousarray(S.T).ravel()), p(b), p(xo))
        hc.hc_ldlt6_perm(p(np.ascontiguousarray(S.T).ravel()), p(b), p(xp))
        hc.hc_ldlt6(p(np.ascontiguousarray(S.T).ravel()), p(b), p(xe))
        d = np.diag(S)
        if np.all(d[:-1] > d[1:]):
            assert xo.tobytes() == xp.tobytes()
        S64, b64 = S.astype(np.float64), b.astype(np.float64)
        res = np.abs(S64 @ xo - b64).max()
        assert res <= 2e-5 * (np.abs(S64).sum(1).max() * np.abs(xo).max() + np.abs(b64).max())
        x64 = np.linalg.solve(S64, b64)
        assert np.abs(xo - x64).max() <= 8 * max(np.abs(xe - x64).max(), 1e-6 * np.abs(x64).max())
    z = np.ones(6, np.float32)
    hc.hc_ldlt6_ordered(p(np.zeros(36, np.float32)), p(np.ones(6, np.float32)), p(z))
    assert np.all(z == 0)
    I6 = np.eye(6, dtype=np.float32) * np.float32(4.0)
    hc.hc_ldlt6_ordered(p(I6.ravel()), p(np.arange(6, dtype=np.float32)), p(z))
    assert np.array_equal(z, np.arange(6, dtype=np.float32) / np.float32(4.0))


def test_small_angle_sincos(hc):
    """sincos_small (the round kernels' sin / cos of a Gauss-Newton step): within 1.5 ulp of the correctly rounded values
    on |x| <= 0.5, exact at 0"""
    import ctypes
    hc.hc_sincos_small.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(-0.5, 0.5, 20000), rng.uniform(-1e-3, 1e-3, 5000), [0.0, 0.5, -0.5, 1e-20, -1e-30]]).astype(np.float32)
    worst_s = worst_c = 0.0
    for x in xs:
        s, c = ctypes.c_float(), ctypes.c_float()
        hc.hc_sincos_small(float(x), ctypes.byref(s), ctypes.byref(c))
        rs, rc = np.sin(np.float64(x)), np.cos(np.float64(x))
        us = float(np.spacing(np.float32(abs(rs)))) if rs != 0 else 1e-45
        worst_s = max(worst_s, abs(float(s.value) - rs) / us)
        worst_c = max(worst_c, abs(float(c.value) - rc) / float(np.spacing(np.float32(rc))))
    assert worst_s <= 1.5 and worst_c <= 1.5, (worst_s, worst_c)
    s, c = ctypes.c_float(), ctypes.c_float()
    hc.hc_sincos_small(0.0, ctypes.byref(s), ctypes.byref(c))
    assert s.value == 0.0 and c.value == 1.0


def test_ldlt2_and_triangulate_point_bit_exact(hc, o32):
    rng = np.random.default_rng(3)
    for _ in range(200):
        d1 = rng.normal(size=3).astype(np.float32); d2 = rng.normal(size=3).astype(np.float32)
        t = rng.normal(size=3).astype(np.float32)
        out = np.zeros(3, np.float32); ref = np.zeros(3, np.float32)
        ok = hc.hc_triangulate_point(p(d1), p(d2), p(t), p(out))
        f = o32._f("triangulate_point"); f.restype = C.c_int
        ok_ref = f(p(d1), p(d2), p(t), p(ref))
        assert ok == ok_ref
        if ok:
            assert out.tobytes() == ref.tobytes()
    m = np.array([4.0, 1.0, 1.0, 9.0], np.float32); rhs = np.array([1.0, 2.0], np.float32); x = np.zeros(2, np.float32)
    hc.hc_ldlt2(p(m), p(rhs), p(x))
    assert x.tobytes() == o32.ldlt_solve(m.reshape(2, 2), rhs).tobytes()


def test_tri_constants_and_v2t(hc, o32, vo):
    rng = np.random.default_rng(4)
    X = vo.synth.random_isometry(rng)
    iK = np.zeros(9, np.float32); iRiK = np.zeros(9, np.float32); t = np.zeros(3, np.float32)
    hc.hc_tri_constants(p(cm(vo.synth.K_REF, 3)), p(cm(X, 4)), p(iK), p(iRiK), p(t))
    ref = np.zeros(9, np.float32)
    o32._f("mat3_inverse")(p(cm(vo.synth.K_REF, 3)), p(ref))
    assert iK.tobytes() == ref.tobytes()
    assert np.allclose(iK.reshape(3, 3).T, np.linalg.inv(vo.synth.K_REF), atol=1e-6)
    assert np.allclose(t, np.linalg.inv(X.astype(np.float64))[:3, 3], atol=1e-6)
    v = rng.uniform(-0.5, 0.5, 6).astype(np.float32)
    T = np.zeros(16, np.float32)
    hc.hc_v2t(p(v), p(T))
    assert T.reshape(4, 4).T.tobytes() == o32.v2t_euler(v).tobytes()


@pytest.mark.parametrize("thr,keep", [(10000.0, 0), (60.0, 0), (60.0, 1)])
def test_exact_mode_arithmetic_is_the_oracles_bit_for_bit(hc, o32, vo, thr, keep):
    """picp_term_exact + sequential sums + picp_update_t<true> (what picp_exact_kernel runs) == ref32, bitwise,
    over many chained rounds: H (with damping), b, chi, inlier count and pose."""
    fp = vo.synth.frame_pair(1200, seed=43, drop=0.05, distractors=5, model_drop=0.05)
    j = o32.join(o32.match(fp["ref_app"], fp["cur_app"]), fp["model_pairs"])
    cam = Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4))
    n_it = 25
    r = o32.picp_solve_raw(cam, fp["model"], fp["cur_pts"], j, n_it, thr, bool(keep))
    tH = np.zeros((n_it, 36), np.float32); tb = np.zeros((n_it, 6), np.float32)
    ts = np.zeros((n_it, 3), np.float32); tT = np.zeros((n_it, 16), np.float32)
    hc.hc_picp_exact(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], p(cm(fp["K"], 3)), p(cm(np.eye(4), 4)),
                     C.c_float(thr), keep, p(fp["model"]), p(fp["cur_pts"]), p(j), len(j), n_it, p(tH), p(tb), p(ts), p(tT))
    assert np.array_equal(tH.reshape(n_it, 6, 6).transpose(0, 2, 1), r["H"])
    assert np.array_equal(tb, r["b"])
    assert np.array_equal(ts, r["stats"])
    assert np.array_equal(tT.reshape(n_it, 4, 4).transpose(0, 2, 1), r["T"])
    if keep == 0 and thr < 100:
        assert 0 < ts[0, 2] < len(j) and ts[0, 1] > 0       # both branches of the chi test taken


def test_point_cloud_update_follows_operator_equal(tmp_path):
    """the facade's map upsert uses a hash index; its key must be the equivalence of the reference's operator==
    (PointCloud.h:56): -0 == +0, an appearance holding a NaN equals nothing (appended every time)"""
    exe = str(tmp_path / "pc_update")
    root = os.path.join(HERE, "..")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "include", "vo"),
                           os.path.join(HERE, "hostcheck", "point_cloud_update.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == ["4", "2"], r.stdout


def test_multi_gpu_rank_bookkeeping_without_a_gpu(tmp_path):
    """include/vo/shard.hpp -- what apps/batch_frames_mgpu.cpp and apps/sequence_mgpu.cpp do between the GPU calls: partition,
    equal padded blocks, per-call slicing, the gathered buffer's layout, own-block / foreign-block checks, global order, and the
    rule that no rank enters a collective unless all do -- with every rank a host thread over fake pose buffers: worlds of
    2, 3 and 8 ranks, 13 and 1601 items, fewer items than ranks, a rank failing at set-up and one failing between two passes
    (the program must come back, not hang)."""
    exe = str(tmp_path / "shard_check")
    root = os.path.join(HERE, "..")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(root, "include"),
                           os.path.join(HERE, "hostcheck", "shard_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("shard_check: all ok"), r.stdout[-3000:]
    assert r.stdout.count("-> ok") >= 40 and "FAILED" not in r.stdout


# ---- host linear algebra of the epipolar initialisation (include/vo/epipolar.hpp, vo/linalg.hpp) ---------------------
@pytest.fixture(scope="module")
def epi():
    so = os.path.join(HERE, "hostcheck", "libvo_epipolar_check.so")
    src = os.path.join(HERE, "hostcheck", "epipolar_check.cpp")
    root = os.path.join(HERE, "..")
    deps = [src, os.path.join(root, "include", "vo", "epipolar.hpp"), os.path.join(root, "include", "vo", "linalg.hpp"),
            os.path.join(root, "visual-odometry_amd", "csrc", "vo_math.h")]
    if not os.path.exists(so) or max(os.path.getmtime(d) for d in deps) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + os.path.join(root, "include"), "-o", so, src])
    return C.CDLL(so)


def _estimate(epi, K, corr, p1, p2):
    K = np.ascontiguousarray(np.asarray(K, np.float32).reshape(3, 3).T)       # column-major
    corr = np.ascontiguousarray(corr, np.int32); p1 = np.ascontiguousarray(p1, np.float32); p2 = np.ascontiguousarray(p2, np.float32)
    X = np.zeros(16, np.float32)
    n_front = epi.hc_estimate_transform(p(K), p(corr), len(corr), p(p1), len(p1), p(p2), len(p2), p(X))
    return X.reshape(4, 4).T.copy(), n_front


def test_epipolar_initialisation_on_the_reference_data(epi, o32):
    """The facade's host code on the first two frames of the reference's data directory with the known association
    (initialization_real_data.cpp): against the ground truth of trajectory.dat and against the oracle's numpy restatement."""
    from oracle import vo_pipeline as vp
    data = os.path.join(HERE, "golden", "example_data", "data")
    r = vp.run_real_init(data, o32)
    X, n_front = _estimate(epi, r["K"], r["corr"], r["p0"], r["p1"])
    assert n_front == len(r["corr"]) == 115
    assert np.abs(X - r["X"]).max() < 2e-5                                  # Jacobi here, LAPACK there; both double inside
    gt = vp.read_gt(os.path.join(data, "trajectory.dat"))
    H = r["H"].astype(np.float64)
    X_gt = np.linalg.inv(H) @ np.linalg.inv(gt[1]) @ gt[0] @ H
    scale = np.linalg.norm(X_gt[:3, 3]) / np.linalg.norm(X[:3, 3])
    assert np.abs(X[:3, :3] - X_gt[:3, :3]).max() < 2e-5 and np.abs(X[:3, 3] * scale - X_gt[:3, 3]).max() < 5e-5


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_epipolar_initialisation_recovers_a_synthetic_motion(epi, o32, vo, seed):
    from oracle import vo_pipeline as vp
    fp = vo.synth.frame_pair(400, seed=seed, noise_px=0.0, max_angle=0.2, max_t=0.5)
    corr = fp["gt_matches"]
    X, n_front = _estimate(epi, fp["K"], corr, fp["ref_pts"], fp["cur_pts"])
    Xo = vp.estimate_transform(o32, fp["K"], corr, fp["ref_pts"], fp["cur_pts"])
    assert n_front == len(corr)
    assert np.abs(X - Xo).max() < 5e-5
    Xg = fp["X_gt"].astype(np.float64)
    t, tg = X[:3, 3] / np.linalg.norm(X[:3, 3]), Xg[:3, 3] / np.linalg.norm(Xg[:3, 3])
    assert np.abs(X[:3, :3] - Xg[:3, :3]).max() < 1e-3 and float(t @ tg) > 0.9999   # pixel coordinates are float32


def test_svd3_of_the_facade(epi):
    rng = np.random.default_rng(5)
    mats = [rng.normal(size=(3, 3)) for _ in range(20)]
    mats.append(np.outer([1.0, 2.0, 3.0], [0.5, -1.0, 2.0]))               # rank 1
    mats.append(np.diag([3.0, 3.0, 0.0]))                                  # repeated singular value, rank 2
    mats.append(np.zeros((3, 3)))
    for rel in (1e-6, 1e-9, 1e-12, 1e-15):                                  # a nearly vanishing singular value: an essential matrix
        Q1, _ = np.linalg.qr(rng.normal(size=(3, 3))); Q2, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        mats.append(Q1 @ np.diag([2.0, 1.5, 2.0 * rel]) @ Q2.T)
    for M in mats:
        M = np.ascontiguousarray(M, np.float64)
        U = np.zeros((3, 3)); V = np.zeros((3, 3)); s = np.zeros(3)
        epi.hc_svd3(p(M), p(U), p(s), p(V))
        assert np.allclose(U @ np.diag(s) @ V.T, M, atol=1e-12)
        assert np.allclose(np.sort(s)[::-1], np.linalg.svd(M, compute_uv=False), atol=1e-12)
        assert s[0] >= s[1] >= s[2] >= 0
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-12) and np.allclose(V.T @ V, np.eye(3), atol=1e-12)   # rank-deficient too


# ---- file readers / writers and the evaluation of the facade (include/vo/files.hpp, vo/evaluation.hpp) -----------------
def test_files_and_evaluation_of_the_facade(tmp_path, o32):
    """The host code a `vo_complete` + `evaluation` user runs around the GPU path, without a GPU: parse the reference's data
    directory, write a trajectory, evaluate it -- against the Python restatement (oracle/vo_pipeline.py) on the same files."""
    import re
    from oracle import vo_pipeline as vp
    root = os.path.join(HERE, "..")
    data = os.path.join(HERE, "golden", "example_data", "data")
    exe = str(tmp_path / "files_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"), os.path.join(HERE, "hostcheck", "files_check.cpp"), "-o", exe])
    res = vp.run_vo_complete(data, rounds=100, o=o32)
    np.savetxt(tmp_path / "poses_in.txt", np.array(res["trajectory"]).reshape(-1, 16), fmt="%.9g")
    np.savetxt(tmp_path / "map.txt", res["map"], fmt="%.9g")
    np.savetxt(tmp_path / "map_appearances.txt", res["map_app"], fmt="%.9g")
    r = subprocess.run([exe, data, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = {line.split()[0] + ("_" + line.split()[1] if line.startswith("meas") else ""): line.split() for line in r.stdout.splitlines()}
    files = sorted(f for f in os.listdir(data) if re.search(r"^meas-\d.*\.dat$", f))
    assert out["files"][1] == "121" and out["files"][3] == files[0] and out["files"][5] == files[-1]
    w10 = np.arange(1, 11, dtype=np.float64)
    for f in (files[0], files[-1]):
        pts, app, ids = vp.read_meas(os.path.join(data, f))
        m = out["meas_" + f]
        assert int(m[3]) == len(ids) == int(m[11])
        assert float(m[5]) == float(ids.sum())
        s_uv = float((pts[:, 0].astype(np.float64) + 2.0 * pts[:, 1]).sum()); s_app = float((app.astype(np.float64) * w10).sum())
        assert abs(float(m[7]) - s_uv) < 1e-9 * abs(s_uv) and abs(float(m[9]) - s_app) < 1e-9 * max(1.0, abs(s_app))
        assert abs(float(m[12]) - (s_uv + s_app)) < 1e-9 * abs(s_uv)                  # the point-cloud reader: same numbers
    world, wapp = vp.read_world(os.path.join(data, "world.dat"))
    s_w = float((world.astype(np.float64) * [1.0, 2.0, 3.0]).sum()); s_wa = float((wapp.astype(np.float64) * w10).sum())
    assert int(out["world"][2]) == len(world) == 1000
    assert abs(float(out["world"][4]) - s_w) < 1e-9 * abs(s_w) and abs(float(out["world"][6]) - s_wa) < 1e-9 * max(1.0, abs(s_wa))
    K, H, ints = vp.read_camera(os.path.join(data, "camera.dat"))
    c = out["camera"]
    assert [int(x) for x in c[2:6]] == [ints["z_near"], ints["z_far"], ints["width"], ints["height"]]
    assert np.array_equal(np.array(c[7:16], np.float32).reshape(3, 3), K) and np.array_equal(np.array(c[17:33], np.float32).reshape(4, 4), H)
    gt = vp.read_gt(os.path.join(data, "trajectory.dat"))
    s_gt = sum(float((np.arange(1, 13).reshape(3, 4) * T[:3]).sum()) for T in gt)
    assert int(out["gt"][2]) == len(gt) == 121 and abs(float(out["gt"][4]) - s_gt) < 1e-5 * abs(s_gt)      # float sin/cos there
    gt_txt = np.loadtxt(tmp_path / "trajectory_gt.txt")
    assert gt_txt.shape == (121, 3) and np.abs(gt_txt - np.array([T[:3, 3] for T in gt])).max() < 1e-5
    # save_trajectory: the composition H C X^-1 C^-1 (float32 there, float64 here)
    ref = vp.robot_trajectory(res["trajectory"], res["H"])
    est = np.loadtxt(tmp_path / "trajectory_est_complete.txt")
    assert est.shape == (121, 3) and np.abs(est - np.array([T[:3, 3] for T in ref])).max() < 2e-4
    data_txt = np.loadtxt(tmp_path / "trajectory_est_data.txt").reshape(121, 4, 3)
    assert np.abs(data_txt[:, 0] - est).max() < 1e-6 and np.abs(data_txt[:, 1:] - np.array([T[:3, :3] for T in ref])).max() < 2e-5
    # evaluate.cpp's numbers
    ev = vp.evaluate(data, res)
    e = out["eval"]
    assert int(e[2]) == 121 and int(e[12]) == ev["matched"]
    assert abs(float(e[4])) < 1e-5                                                     # float32 noise, like the README's 5.3e-6
    assert abs(float(e[6]) - ev["median_ratio_inv"]) < 1e-4 * ev["median_ratio_inv"]
    assert abs(float(e[8]) - ev["rmse_position"]) < 2e-3 * ev["rmse_position"]
    assert abs(float(e[10]) - ev["rmse_map"]) < 2e-3 * ev["rmse_map"]
    perf = np.loadtxt(tmp_path / "out_performance.txt")
    assert perf.shape == (120, 2)
