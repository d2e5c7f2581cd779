"""Parity of the HIP path (through the C ABI) with the CPU oracle.

Bars: indices, counts and survivor order exact; H, b per iteration
within 2e-3 of ref32 (the noise of the *reference's* sequential float sum) and
1e-5 of ref64; pose after K rounds within 1e-4; triangulated points within
1e-4*max(1,|p|).  Every test goes through libvo_hip.so; none can pass without it.
"""
import os

import numpy as np
import pytest

from conftest import rel_err
from oracle.oracle import Camera as OCam

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    d = dict(np.load(os.path.join(GOLD, name)))
    r, c, zn, zf = d["cam_ints"].tolist()
    d.update(rows=r, cols=c, z_near=zn, z_far=zf)
    return d


def gpu_trace(vo, ctx, fp, world, meas, corr, n_iters, thr, keep, T0=None):
    cam = vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"],
                    np.eye(4) if T0 is None else T0, ctx=ctx)
    s = vo.PICPSolver(ctx)
    s.setKernelThreshold(thr)
    s.init(cam, world, meas)
    out = dict(H=[], b=[], stats=[], T=[])
    for _ in range(n_iters):
        assert s.oneRound(corr, keep) is True
        H, b = s.system()
        out["H"].append(H - np.eye(6, dtype=np.float32))
        out["b"].append(b)
        out["stats"].append((s.chiInliers(), s.chiOutliers(), s.numInliers()))
        out["T"].append(s.camera().worldInCameraPose())
    s.close()
    return {k: np.array(v) for k, v in out.items()}


@pytest.mark.parametrize("name", ["frame64.npz", "frame1000.npz"])
def test_golden_frame_pipeline(vo, ctx, name):
    g = gold(name)
    m = vo.compute_correspondences_images(g["ref_app"], g["cur_app"], ctx=ctx)
    assert np.array_equal(m, g["exp_match"])
    j = vo.extract_correspondences_world(m, g["model_pairs"], ctx=ctx)
    assert np.array_equal(j, g["exp_join"])
    for tag, thr, keep in (("a", 10000.0, False), ("b", 60.0, False), ("c", 60.0, True)):
        n_it = len(g[f"picp_{tag}_T32"])
        t = gpu_trace(vo, ctx, g, g["model"], g["cur_pts"], j, n_it, thr, keep)
        # round 0 starts from the same pose: compare the normal equations themselves
        assert rel_err(t["H"][0], g[f"picp_{tag}_H64"][0]) < 1e-5
        assert rel_err(t["b"][0], g[f"picp_{tag}_b64"][0]) < 1e-5
        assert rel_err(t["H"][0], g[f"picp_{tag}_H32"][0]) < 2e-3
        assert np.array_equal(t["stats"][:, 2], g[f"picp_{tag}_stats32"][:, 2])       # inlier counts, every round
        assert np.allclose(t["stats"][:, :2], g[f"picp_{tag}_stats64"][:, :2], rtol=1e-4, atol=1e-3)
        assert np.abs(t["T"] - g[f"picp_{tag}_T32"]).max() < 1e-4
        assert np.abs(t["T"] - g[f"picp_{tag}_T64"]).max() < 1e-4
    T = g["picp_a_T32"][-1]
    xyz, pairs, app = vo.triangulate_points(g["K"], T, m, g["ref_pts"], g["cur_pts"], g["cur_app"], ctx=ctx)
    assert np.array_equal(pairs, g["exp_tri_pairs"])
    assert np.array_equal(app, g["exp_tri_app"])
    assert np.array_equal(xyz, g["exp_tri_xyz"])            # same pose in, reference operation order, no FMA: bit-exact
    xt = vo.transform_points(T, g["model"], ctx=ctx)
    assert np.array_equal(xt, g["exp_transform"])           # same operation order, no FMA: bit-exact


def test_picp_test_scenarios(vo, ctx):
    g = gold("picp_test1009.npz")
    cam = vo.Camera(g["rows"], g["cols"], g["z_near"], g["z_far"], g["K"], g["X_gt"], ctx=ctx)
    uv, n_in = cam.projectPoints(g["world"], keep_indices=True)
    assert n_in == int(g["exp_proj_inside"]) and np.array_equal(uv, g["exp_proj_keep"])   # bit-exact projection
    uv2, n_in2 = cam.projectPoints(g["world"], keep_indices=False)
    assert n_in2 == n_in and np.array_equal(uv2, g["exp_proj_compact"])
    ok, one = cam.projectPoint(g["world"][int(g["corr"][0, 1])])
    assert ok and np.array_equal(one, g["exp_proj_keep"][int(g["corr"][0, 1])])
    t = gpu_trace(vo, ctx, g, g["world"], g["cur_pts"], g["corr"], 100, 10000.0, False)
    assert np.array_equal(t["stats"][:, 2], g["picp_stats32"][:, 2])
    assert np.abs(t["T"] - g["picp_T32"]).max() < 1e-4
    assert np.abs(t["T"][-1] - g["X_gt"]).max() < 1e-3
    g0 = gold("picp_test1000.npz")      # zero inliers: H = I, b = 0, the pose must not move
    t0 = gpu_trace(vo, ctx, g0, g0["world"], g0["cur_pts"], g0["corr"], 5, 10000.0, False)
    assert np.all(t0["stats"][:, 2] == 0) and np.all(t0["H"] == 0) and np.all(t0["b"] == 0)
    assert np.array_equal(t0["T"][-1], np.eye(4, dtype=np.float32))
    assert np.allclose(t0["stats"][:, 1], g0["picp_stats32"][:5, 1], rtol=1e-5)


def test_solve_equals_repeated_one_round(vo, ctx, o32):
    """The fused n-iteration entry point (graph of launches) must give bit-identical
    results to n oneRound calls, run-to-run as well (deterministic reductions)."""
    fp = vo.synth.frame_pair(5000, seed=77, drop=0.05, distractors=20, model_drop=0.05)
    m = vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=ctx)
    j = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
    cam = vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx)
    res = []
    for mode in ("rounds", "solve", "solve"):
        s = vo.PICPSolver(ctx)
        s.setKernelThreshold(10000.0)
        s.init(cam, fp["model"], fp["cur_pts"])
        if mode == "rounds":
            for _ in range(12):
                s.oneRound(j, False)
        else:
            s.solve(j, False, 12)
        res.append((s.camera().worldInCameraPose().tobytes(), s.system()[0].tobytes(), s.numInliers()))
        s.close()
    assert res[0] == res[1] == res[2]
    r = o32.picp_solve(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)),
                       fp["model"], fp["cur_pts"], j, 12, 10000.0, False, trace=False)
    T = np.frombuffer(res[0][0], dtype=np.float32).reshape(4, 4)
    assert np.abs(T - r["T"]).max() < 1e-4
    assert res[0][2] == r["num_inliers"]


def test_edge_cases(vo, ctx):
    e10 = np.zeros((0, 10), np.float32)
    one = np.zeros((1, 10), np.float32)
    assert len(vo.compute_correspondences_images(e10, one, ctx=ctx)) == 0
    assert len(vo.compute_correspondences_images(one, e10, ctx=ctx)) == 0
    assert vo.compute_correspondences_images(one, one, ctx=ctx).tolist() == [[0, 0]]
    # strict radius and lowest-index tie (brute_force_search.h:35)
    base = np.zeros((3, 10), np.float32); base[1, 0] = 5; base[2, 0] = -5
    q = np.zeros((1, 10), np.float32); q[0, 1] = np.float32(0.1)
    assert len(vo.compute_correspondences_images(base, q, ctx=ctx)) == 0
    q[0, 1] = np.float32(0.0999)
    assert vo.compute_correspondences_images(base, q, ctx=ctx).tolist() == [[0, 0]]
    dup = np.concatenate([base, base[:1]])
    assert vo.compute_correspondences_images(dup, q, ctx=ctx).tolist() == [[0, 0]]
    assert vo.compute_correspondences_images(q, dup, ctx=ctx).tolist() == [[0, 0]]
    # join: first partner wins, missing partner dropped, empty inputs
    img = np.array([[5, 0], [2, 1], [9, 2], [2, 3]], np.int32)
    world = np.array([[2, 70], [5, 71], [2, 72]], np.int32)
    assert vo.extract_correspondences_world(img, world, ctx=ctx).tolist() == [[0, 71], [1, 70], [3, 70]]
    assert len(vo.extract_correspondences_world(np.zeros((0, 2), np.int32), world, ctx=ctx)) == 0
    assert len(vo.extract_correspondences_world(img, np.zeros((0, 2), np.int32), ctx=ctx)) == 0
    # empty projection / triangulation / transform
    cam = vo.Camera(480, 640, 0, 10, vo.synth.K_REF, np.eye(4), ctx=ctx)
    uv, n = cam.projectPoints(np.zeros((0, 3), np.float32))
    assert len(uv) == 0 and n == 0
    x, p, _ = vo.triangulate_points(vo.synth.K_REF, np.eye(4), np.zeros((0, 2), np.int32),
                                    np.zeros((1, 2), np.float32), np.zeros((1, 2), np.float32), ctx=ctx)
    assert len(x) == 0 and len(p) == 0
    assert len(vo.transform_points(np.eye(4), np.zeros((0, 3), np.float32), ctx=ctx)) == 0
    # a solver with zero correspondences: H = I, pose unchanged
    s = vo.PICPSolver(ctx)
    s.init(cam, np.zeros((1, 3), np.float32), np.zeros((1, 2), np.float32))
    s.oneRound(np.zeros((0, 2), np.int32), False)
    assert np.array_equal(s.camera().worldInCameraPose(), np.eye(4, dtype=np.float32)) and s.numInliers() == 0
    # out-of-range correspondence index is reported, not dereferenced
    s.oneRound(np.array([[0, 5]], np.int32), False)
    with pytest.raises(vo.VoError):
        s.numInliers()
    s.close()


def test_gates_match_oracle_bitwise(vo, ctx, o32):
    rng = np.random.default_rng(5)
    T = vo.synth.random_isometry(rng, 0.3, 0.5)
    pts = vo.synth.random_points3d(rng, 20000) * np.float32([0.3, 0.3, 4.0])
    cam = vo.Camera(480, 640, 0, 10, vo.synth.K_REF, T, ctx=ctx)
    uv, n_in = cam.projectPoints(pts, keep_indices=True)
    uv_o, n_o = o32.project_points(OCam(480, 640, 0, 10, vo.synth.K_REF, T), pts, keep_indices=True)
    assert n_in == n_o and uv.tobytes() == uv_o.tobytes()
    uvc, _ = cam.projectPoints(pts, keep_indices=False)
    uvc_o, _ = o32.project_points(OCam(480, 640, 0, 10, vo.synth.K_REF, T), pts, keep_indices=False)
    assert uvc.tobytes() == uvc_o.tobytes() and 0 < len(uvc) < len(pts)


def test_matcher_variants_agree(vo, o32):
    """Full scan (mode 1), bucket-pruned scan (mode 2), cell-hash search (mode 3) and both of the latter behind the
    exact-duplicate pass (modes 4, 5) must return the oracle's pairs on every input, including duplicates (ties -> lowest
    index), degenerate spreads and tiny sets."""
    import ctypes as C
    c1, c2, c3, c4, c5 = (vo.Context(0) for _ in range(5))
    for k, c in enumerate((c1, c2, c3, c4, c5)):
        assert c.lib.vo_match_set_mode(c.h, k + 1) == 0
    assert c1.lib.vo_match_set_mode(c1.h, 7) != 0
    rng = np.random.default_rng(17)
    cases = []
    for n, seed, kw in ((50, 1, dict(drop=0.2, distractors=3)), (1500, 3, dict(drop=0.3, distractors=9)),
                        (6000, 4, dict(drop=0.05, distractors=200))):
        fp = vo.synth.frame_pair(n, seed=seed, **kw)
        cases.append((fp["ref_app"], fp["cur_app"]))
    base = rng.uniform(-1, 1, (400, 10)).astype(np.float32)
    cases.append((np.concatenate([base, base, base[::-1]]), base))                 # exact duplicates: ties
    cases.append((np.concatenate([base + rng.normal(0, 0.01, base.shape).astype(np.float32) for _ in range(4)]), base))
    flat = base.copy(); flat[:, :] = flat[:, :1] * 0 + rng.uniform(-1e-3, 1e-3, base.shape).astype(np.float32)
    cases.append((flat, flat[:100] + np.float32(1e-4)))                             # everything within the radius
    const = np.zeros((300, 10), np.float32)
    cases.append((const, const[:7]))                                               # zero spread in every dimension
    one_dim = base.copy(); one_dim[:, 1:] = 0
    cases.append((one_dim, one_dim[:50] + np.float32(0.05)))
    # lattice data (multiples of 1/32): many distinct points at EXACTLY equal distance, and distances
    # that sit exactly on and one step beside radius^2 -- the fused prefilter must not lose or add any
    lat = (rng.integers(-3, 4, (2500, 10)) / 32.0).astype(np.float32)
    cases.append((lat, lat[:600] + (rng.integers(-1, 2, (600, 10)) / 32.0).astype(np.float32)))
    edge_q = np.zeros((64, 10), np.float32)
    edge_t = np.zeros((4, 10), np.float32)
    edge_t[0, 0] = np.float32(0.1); edge_t[1, 0] = np.nextafter(np.float32(0.1), np.float32(0))
    edge_t[2, :4] = np.float32(0.05); edge_t[3, :4] = np.nextafter(np.float32(0.05), np.float32(0))
    cases.append((np.concatenate([edge_t, lat[:300] + 2]), edge_q))
    for a, b in cases:
        for x, y in ((a, b), (b, a)):
            exp = o32.match(x, y)
            assert np.array_equal(vo.compute_correspondences_images(x, y, ctx=c1), exp)
            assert np.array_equal(vo.compute_correspondences_images(x, y, ctx=c2), exp)
            assert np.array_equal(vo.compute_correspondences_images(x, y, ctx=c3), exp)
            assert np.array_equal(vo.compute_correspondences_images(x, y, ctx=c4), exp)
            assert np.array_equal(vo.compute_correspondences_images(x, y, ctx=c5), exp)
    for c in (c1, c2, c3, c4, c5): c.close()


def test_matcher_sizes_and_branches(vo, ctx, o32):
    """Both tree/query role assignments, sizes that are not multiples of any
    tile, and a query set larger than one workgroup."""
    for n, seed, kw in ((37, 1, dict(drop=0.2, distractors=3)), (700, 2, dict(drop=0.1, distractors=40)),
                        (1500, 3, dict(drop=0.3, distractors=0)), (2600, 4, dict(drop=0.0, distractors=100))):
        fp = vo.synth.frame_pair(n, seed=seed, **kw)
        for a, b in ((fp["ref_app"], fp["cur_app"]), (fp["cur_app"], fp["ref_app"])):
            assert np.array_equal(vo.compute_correspondences_images(a, b, ctx=ctx), o32.match(a, b))
    # near-duplicate appearances (within the radius, distinct distances): argmin must agree
    rng = np.random.default_rng(9)
    base = rng.uniform(-1, 1, (300, 10)).astype(np.float32)
    tree = np.concatenate([base + rng.normal(0, 0.01, base.shape).astype(np.float32) for _ in range(4)])
    m_g = vo.compute_correspondences_images(tree, base, ctx=ctx)
    m_o = o32.match(tree, base)
    assert np.array_equal(m_g, m_o) and len(m_o) > 250


def test_radius_search_is_the_oracles_full_search(vo, ctx, o32):
    """vo_radius_search = TreeNode_::fullSearch for every query: the exact set of tree points inside the ball,
    against the oracle's kd-tree traversal and plain double loop."""
    rng = np.random.default_rng(12)
    base = rng.uniform(-1, 1, (700, 10)).astype(np.float32)
    tree = np.concatenate([base + rng.normal(0, 0.02, base.shape).astype(np.float32) for _ in range(6)])    # long lists
    lat = (rng.integers(-2, 3, (900, 10)) / 32.0).astype(np.float32)                                        # boundary distances
    fp = vo.synth.frame_pair(5000, seed=88, drop=0.2, distractors=50)                                        # mostly 0/1 hits
    for t, q, r in ((tree, base, 0.1), (base, tree, 0.1), (lat, lat[:150], 0.0625), (lat[:7], lat, 0.09),
                    (fp["ref_app"], fp["cur_app"], 0.1), (np.zeros((40, 10), np.float32), np.zeros((3, 10), np.float32), 0.1)):
        got = vo.radius_search(t, q, r, ctx=ctx)
        exp = o32.radius_search(t, q, r)
        assert len(got) == len(exp) == len(q)
        assert all(np.array_equal(a, b) for a, b in zip(got, exp)), (len(t), len(q), r)
    assert sum(len(x) for x in vo.radius_search(tree, base, 0.1, ctx=ctx)) > 3000
    # empty sets and the capacity protocol of the C entry point
    assert [len(x) for x in vo.radius_search(np.zeros((0, 10), np.float32), base[:4], ctx=ctx)] == [0, 0, 0, 0]
    import ctypes as C
    off = np.zeros(len(base) + 1, np.int32); idx = np.zeros(8, np.int32); n_total = C.c_int()
    rc = ctx.lib.vo_radius_search(ctx.h, tree.ctypes.data_as(C.c_void_p), C.c_int(len(tree)), base.ctypes.data_as(C.c_void_p),
                                  C.c_int(len(base)), C.c_float(0.1), off.ctypes.data_as(C.c_void_p),
                                  idx.ctypes.data_as(C.c_void_p), C.c_int(8), C.byref(n_total))
    assert rc == -1 and n_total.value > 3000 and off[-1] == n_total.value
