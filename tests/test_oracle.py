"""CPU tests of the oracle itself: ref32 vs ref64 vs an independent numpy
restatement, analytic known answers, the frozen golden fixtures, and the
reference's edge-case semantics."""
import os

import numpy as np
import pytest

import np_restatement as npr
from conftest import rel_err
from oracle.oracle import Camera

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cam(fp, T=None):
    return Camera(int(fp["rows"]), int(fp["cols"]), int(fp["z_near"]), int(fp["z_far"]), fp["K"],
                  np.eye(4) if T is None else T)


def _gold(name):
    d = dict(np.load(os.path.join(GOLD, name)))
    r, c, zn, zf = d["cam_ints"].tolist()
    d.update(rows=r, cols=c, z_near=zn, z_far=zf)
    return d


@pytest.mark.parametrize("name", ["frame64.npz", "frame1000.npz"])
def test_golden_frames_are_reproduced(o32, o64, name):
    g = _gold(name)
    m = o32.match(g["ref_app"], g["cur_app"])
    assert np.array_equal(m, g["exp_match"])
    j = o32.join(m, g["model_pairs"])
    assert np.array_equal(j, g["exp_join"])
    assert np.array_equal(o32.join(m, g["model_pairs"], linear=True), j)
    for tag, thr, keep in (("a", 10000.0, False), ("b", 60.0, False), ("c", 60.0, True)):
        r = o32.picp_solve(_cam(g), g["model"], g["cur_pts"], j, len(g[f"picp_{tag}_T32"]), thr, keep)
        assert np.array_equal(r["T_trace"], g[f"picp_{tag}_T32"])      # bit-stable restatement
        assert np.array_equal(r["stats"], g[f"picp_{tag}_stats32"])
        r64 = o64.picp_solve(_cam(g), g["model"], g["cur_pts"], j, len(g[f"picp_{tag}_T64"]), thr, keep)
        assert np.allclose(r64["T_trace"], g[f"picp_{tag}_T64"], rtol=0, atol=1e-12)
    T = g["picp_a_T32"][-1]
    xyz, pairs, app = o32.triangulate(g["K"], T, m, g["ref_pts"], g["cur_pts"], g["cur_app"])
    assert np.array_equal(pairs, g["exp_tri_pairs"]) and np.array_equal(xyz, g["exp_tri_xyz"])
    assert np.array_equal(app, g["exp_tri_app"])
    assert np.array_equal(o32.transform_points(T, g["model"]), g["exp_transform"])


def test_matcher_equals_ground_truth_and_numpy(o32, vo):
    for seed, kw in ((11, dict(drop=0.1, distractors=30)), (12, dict(drop=0.2, distractors=7)), (13, dict())):
        fp = vo.synth.frame_pair(400, seed=seed, **kw)
        m = o32.match(fp["ref_app"], fp["cur_app"])
        gt = fp["gt_matches"]
        tree_is_ref = len(fp["ref_app"]) >= len(fp["cur_app"])
        gt = gt[np.argsort(gt[:, 1 if tree_is_ref else 0], kind="stable")]   # query order
        assert np.array_equal(m, gt)
        assert np.array_equal(m, npr.match(fp["ref_app"], fp["cur_app"]))
    # both branches of vo_complete.cpp:20-33 are exercised
    sizes = [(len(vo.synth.frame_pair(400, seed=s, drop=0.1, distractors=30)["ref_app"]),
              len(vo.synth.frame_pair(400, seed=s, drop=0.1, distractors=30)["cur_app"])) for s in (11, 14, 15, 16)]
    assert any(a > b for a, b in sizes) or any(a < b for a, b in sizes)


def test_reference_kdtree_equals_exact_search(o32, vo):
    """The reference's PCA kd-tree + bestMatchFull (restated in oracle/vo_kdtree.c) returns exactly the
    nearest neighbour within the radius, i.e. the definition the oracle and the kernels implement."""
    for n, seed, kw in ((60, 1, dict(drop=0.2, distractors=5)), (900, 2, dict(drop=0.1, distractors=60)),
                        (5000, 3, dict(drop=0.05, distractors=300))):
        fp = vo.synth.frame_pair(n, seed=seed, **kw)
        for a, b in ((fp["ref_app"], fp["cur_app"]), (fp["cur_app"], fp["ref_app"])):
            assert np.array_equal(o32.match_kdtree(a, b), o32.match(a, b))
    rng = np.random.default_rng(4)                      # noisy near-duplicates: several candidates inside the radius
    base = rng.uniform(-1, 1, (400, 10)).astype(np.float32)
    tree = np.concatenate([base + rng.normal(0, 0.01, base.shape).astype(np.float32) for _ in range(4)])
    assert np.array_equal(o32.match_kdtree(tree, base), o32.match(tree, base))
    same = np.zeros((50, 10), np.float32)               # would recurse forever in the reference
    assert len(o32.match_kdtree(same, same[:3])) == 3


def test_matcher_radius_strict_and_ties(o32):
    base = np.zeros((3, 10), dtype=np.float32)
    base[1, 0] = 5.0
    base[2, 0] = -5.0
    q = np.zeros((1, 10), dtype=np.float32)
    q[0, 1] = np.float32(0.1)        # d2 = 0.1f*0.1f exactly: strict '<' rejects  (brute_force_search.h:35)
    assert len(o32.match(base, q)) == 0
    q[0, 1] = np.float32(0.0999)
    assert o32.match(base, q).tolist() == [[0, 0]]
    dup = np.concatenate([base, base[:1]])          # exact tie between tree points 0 and 3 -> lowest index
    assert o32.match(dup, q).tolist() == [[0, 0]]
    # roles: smaller set queries, pairs always (a1 idx, a2 idx)  (vo_complete.cpp:40-43)
    assert o32.match(q, dup).tolist() == [[0, 0]]
    assert len(o32.match(np.zeros((0, 10), np.float32), q)) == 0


def test_join_semantics(o32):
    img = np.array([[5, 0], [2, 1], [9, 2], [2, 3]], dtype=np.int32)
    world = np.array([[2, 70], [5, 71], [2, 72]], dtype=np.int32)     # duplicate ref 2: FIRST wins
    exp = [[0, 71], [1, 70], [3, 70]]                                   # ref 9 has no partner: dropped
    assert o32.join(img, world).tolist() == exp
    assert o32.join(img, world, linear=True).tolist() == exp
    assert len(o32.join(np.zeros((0, 2), np.int32), world)) == 0
    assert len(o32.join(img, np.zeros((0, 2), np.int32))) == 0
    rng = np.random.default_rng(3)
    a = rng.integers(0, 50, (300, 2)).astype(np.int32)
    b = rng.integers(0, 50, (200, 2)).astype(np.int32)
    assert np.array_equal(o32.join(a, b), o32.join(a, b, linear=True))


def test_picp_ref32_ref64_numpy_agree(o32, o64, vo):
    fp = vo.synth.frame_pair(2000, seed=21, drop=0.05, distractors=10, model_drop=0.05)
    j = o32.join(o32.match(fp["ref_app"], fp["cur_app"]), fp["model_pairs"])
    for thr, keep in ((10000.0, False), (50.0, False), (50.0, True)):
        r32 = o32.picp_solve(_cam(fp), fp["model"], fp["cur_pts"], j, 8, thr, keep)
        r64 = o64.picp_solve(_cam(fp), fp["model"], fp["cur_pts"], j, 8, thr, keep)
        Tn, hist = npr.solve(fp["K"], np.eye(4), fp["model"], fp["cur_pts"], j, 8, thr, keep,
                             fp["rows"], fp["cols"], fp["z_near"], fp["z_far"])
        # independent float64 restatement == ref64 (pivoting vs LAPACK: tiny differences)
        assert rel_err(r64["H"][0], hist[0][0]) < 1e-12
        assert rel_err(r64["b"][0], hist[0][1]) < 1e-11
        assert int(r64["stats"][0, 2]) == hist[0][4]
        assert np.abs(r64["T"] - Tn).max() < 1e-9
        # ref32 is ref64 up to float32 sequential-sum noise
        assert rel_err(r32["H"][0], r64["H"][0]) < 2e-5
        assert np.abs(r32["T"] - r64["T"]).max() < 2e-5
        assert int(r32["stats"][0, 2]) == int(r64["stats"][0, 2])
    # exercised both inliers and outliers
    assert 0 < r32["stats"][0, 2] < len(j)


def test_picp_converges_to_ground_truth(o32, vo):
    fp = vo.synth.frame_pair(3000, seed=22, noise_px=0.0)
    j = o32.join(o32.match(fp["ref_app"], fp["cur_app"]), fp["model_pairs"])
    r = o32.picp_solve(_cam(fp), fp["model"], fp["cur_pts"], j, 30, 10000.0, False, trace=False)
    assert np.abs(r["T"] - fp["X_gt"]).max() < 2e-4
    assert r["num_inliers"] == len(j) == 3000


def test_picp_test_scenes(o32):
    g = _gold("picp_test1009.npz")
    r = o32.picp_solve(_cam(g), g["world"], g["cur_pts"], g["corr"], 1000, 10000.0, trace=False)
    assert np.abs(r["T"] - g["X_gt"]).max() < 1e-5               # picp_solver_test.cpp scenario
    g0 = _gold("picp_test1000.npz")                                # all outliers: H = I, b = 0, pose frozen
    r0 = o32.picp_solve(_cam(g0), g0["world"], g0["cur_pts"], g0["corr"], 5, 10000.0)
    assert r0["num_inliers"] == 0 and np.array_equal(r0["T"], np.eye(4, dtype=np.float32))
    assert np.all(r0["H"] == 0) and r0["chi_outliers"] > 0


def test_gates_are_literal(o32):
    """camera.h:28-35: inclusive int depth bounds, u vs cols-1, v vs rows-1."""
    K = np.array([[128, 0, 64], [0, 128, 32], [0, 0, 1]], dtype=np.float32)   # exactly representable edges
    cam = Camera(65, 129, 1, 4, K, np.eye(4))          # rows=65, cols=129
    pts = np.array([[0, 0, 1.0], [0, 0, 4.0], [0, 0, 0.999], [0, 0, 4.001],
                    [0.5, 0, 1.0], [0.5001, 0, 1.0], [-0.5, 0, 1.0], [-0.5001, 0, 1.0],
                    [0, 0.25, 1.0], [0, 0.2501, 1.0], [0, -0.25, 1.0], [0, -0.2501, 1.0]], dtype=np.float32)
    uv, n_in = o32.project_points(cam, pts, keep_indices=True)
    ok = (uv[:, 0] >= 0).tolist()
    assert ok == [True, True, False, False, True, False, True, False, True, False, True, False]
    assert n_in == 6 and len(uv) == 12 and np.all(uv[~np.array(ok)] == -1)
    uv2, n2 = o32.project_points(cam, pts, keep_indices=False)   # camera.cpp:31: stable compaction
    assert n2 == 6 and np.array_equal(uv2, uv[np.array(ok)])
    e, n0 = o32.project_points(cam, np.zeros((0, 3), np.float32))
    assert len(e) == 0 and n0 == 0


def test_ldlt_matches_numpy_and_pivots(o32, o64):
    rng = np.random.default_rng(5)
    for n in (2, 6):
        for _ in range(20):
            A = rng.normal(size=(n + 3, n))
            S = A.T @ A + np.diag(rng.uniform(0, 50, n))
            b = rng.normal(size=n)
            x = o64.ldlt_solve(S, b)
            assert np.allclose(x, np.linalg.solve(S, b), rtol=1e-9, atol=1e-12)
            x32 = o32.ldlt_solve(S, b)
            assert np.allclose(x32, x, rtol=2e-3, atol=1e-5)
    assert np.all(o32.ldlt_solve(np.zeros((6, 6)), np.ones(6)) == 0)      # zero matrix: D^+ = 0
    P = np.diag([1.0, 0.0, 4.0, 0.0, 9.0, 2.0])                             # semi-definite: pseudo-inverse
    assert np.allclose(o64.ldlt_solve(P, np.ones(6)), [1, 0, 0.25, 0, 1 / 9, 0.5])


def test_v2t_and_triangulation_against_numpy(o32, o64, vo):
    v = np.array([0.1, -0.2, 0.3, 0.4, -0.5, 0.6])
    assert np.allclose(o64.v2t_euler(v), npr.v2t_euler(v), atol=1e-15)
    assert np.allclose(o32.v2t_euler(v), npr.v2t_euler(v), atol=1e-6)
    fp = vo.synth.frame_pair(500, seed=31, drop=0.1, distractors=10)
    m = o32.match(fp["ref_app"], fp["cur_app"])
    xyz, pairs, _ = o64.triangulate(fp["K"], fp["X_gt"], m, fp["ref_pts"], fp["cur_pts"])
    xn, pn = npr.triangulate(fp["K"], fp["X_gt"], m, fp["ref_pts"], fp["cur_pts"])
    assert np.array_equal(pairs, pn) and np.allclose(xyz, xn, rtol=1e-9, atol=1e-9)
    assert 0 < len(pairs) < len(m)                      # the cheirality reject (utils.cpp:41) fires
    x32, p32, a32 = o32.triangulate(fp["K"], fp["X_gt"], m, fp["ref_pts"], fp["cur_pts"], fp["cur_app"])
    assert np.array_equal(p32[:, 1], np.arange(len(p32)))            # (idx_second, dense slot)
    assert np.array_equal(a32, fp["cur_app"][p32[:, 0]])               # utils.cpp:127
    # noise-free pair: triangulation recovers the model points (in the reference frame)
    fq = vo.synth.frame_pair(300, seed=32, noise_px=0.0)
    mq = o32.match(fq["ref_app"], fq["cur_app"])
    xq, pq, _ = o64.triangulate(fq["K"], fq["X_gt"], mq, fq["ref_pts"], fq["cur_pts"])
    model_of_ref = dict(fq["model_pairs"].tolist())
    ref_of_cur = {c: r for r, c in mq.tolist()}
    truth = np.array([fq["model"][model_of_ref[ref_of_cur[c]]] for c in pq[:, 0]])
    assert np.abs(xq - truth).max() < 5e-3


def test_all_cores_baseline_is_the_same_solver(o32, vo):
    """vo32_picp_solve_mt (bench.py's all-cores CPU baseline): one thread is bit-identical to the
    sequential restatement; several threads only change the summation order (2e-6 abs on the pose)."""
    fp = vo.synth.frame_pair(3000, seed=77, noise_px=0.5)
    cam = _cam(fp)
    corr = o32.join(o32.match(fp["ref_app"], fp["cur_app"]), fp["model_pairs"])
    seq = o32.picp_solve(cam, fp["model"], fp["cur_pts"], corr, 8, 10000.0, False, trace=True)
    one = o32.picp_solve_mt(cam, fp["model"], fp["cur_pts"], corr, 8, 1, 10000.0)
    assert one["threads"] == 1 and np.array_equal(one["T"], seq["T"])
    assert np.array_equal(one["b"], seq["b"][-1]) and one["num_inliers"] == seq["num_inliers"]
    many = o32.picp_solve_mt(cam, fp["model"], fp["cur_pts"], corr, 8, 3, 10000.0)
    assert many["threads"] == 3 and many["num_inliers"] == seq["num_inliers"]
    assert np.abs(many["T"] - seq["T"]).max() < 2e-6


def test_radius_search_kdtree_equals_brute_force(o32):
    """oracle-side fullSearch (eigen_kdtree.h:56-71): the PCA kd-tree traversal returns exactly the points of the
    plain double loop, on clustered data where the lists are long and on lattice data with distances on the boundary"""
    rng = np.random.default_rng(4)
    base = rng.uniform(-1, 1, (300, 10)).astype(np.float32)
    tree = np.concatenate([base + rng.normal(0, 0.02, base.shape).astype(np.float32) for _ in range(5)])
    a = o32.radius_search(tree, base, 0.1)
    b = o32.radius_search(tree, base, 0.1, brute=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and sum(len(x) for x in a) > 1000
    lat = (rng.integers(-2, 3, (800, 10)) / 32.0).astype(np.float32)
    a = o32.radius_search(lat, lat[:100], 0.0625)                  # neighbours at exactly the radius are excluded (strict <)
    b = o32.radius_search(lat, lat[:100], 0.0625, brute=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert [len(x) for x in o32.radius_search(np.zeros((0, 10), np.float32), base[:3])] == [0, 0, 0]


@pytest.mark.parametrize("seed,angle,t", [(41, 0.5, 0.8), (42, 0.3, 1.0), (43, 0.7, 0.4)])
def test_general_motions_have_their_analytic_answers(o32, o64, vo, seed, angle, t):
    """Large rotations / translations (the other synthetic tests move the camera by 0.05 rad / 0.1): triangulation recovers
    the generating points, the solver converges to the generating pose from the identity, the eight-point initialisation
    recovers rotation and translation direction -- formulas that are only right for small motions would fail here."""
    from oracle import vo_pipeline as vp
    fp = vo.synth.frame_pair(400, seed=seed, noise_px=0.0, max_angle=angle, max_t=t)
    m = o32.match(fp["ref_app"], fp["cur_app"])
    assert np.array_equal(m, fp["gt_matches"])
    Xg = fp["X_gt"].astype(np.float64)
    assert np.abs(Xg[:3, :3] - np.eye(3)).max() > 0.05                     # a motion worth the name
    for o in (o64, o32):
        xyz, pairs, _ = o.triangulate(fp["K"], fp["X_gt"], m, fp["ref_pts"], fp["cur_pts"])
        model_of_ref = dict(fp["model_pairs"].tolist())
        ref_of_cur = {c: r for r, c in m.tolist()}
        truth = np.array([fp["model"][model_of_ref[ref_of_cur[c]]] for c in pairs[:, 0]])
        e = np.abs(xyz - truth).max(axis=1)
        assert len(pairs) == len(m)
        if o is o64:
            assert e.max() < 5e-3                                          # the pixel coordinates are float32
        else:                                                              # float32 depth from a short baseline: far points are loose
            assert np.median(e) < 2e-3 and np.quantile(e, 0.95) < 0.1
    j = o32.join(m, fp["model_pairs"])
    r = o32.picp_solve(Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)), fp["model"], fp["cur_pts"], j, 60,
                       10000.0, False, trace=False)
    assert np.abs(r["T"] - Xg).max() < 1e-4 and r["num_inliers"] == len(j)
    X = vp.estimate_transform(o32, fp["K"], m, fp["ref_pts"], fp["cur_pts"]).astype(np.float64)
    d, dg = X[:3, 3] / np.linalg.norm(X[:3, 3]), Xg[:3, 3] / np.linalg.norm(Xg[:3, 3])
    assert np.abs(X[:3, :3] - Xg[:3, :3]).max() < 1e-4 and float(d @ dg) > 1 - 1e-7
