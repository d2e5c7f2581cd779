"""Every operator of the path against the oracle at set sizes around the edges the kernels tile on: wave (64), workgroup
(256), float4 groups (4), LDS tiles (128), compaction blocks, the matcher's mode switch-overs.  Integer outputs, survivor
order, projection / transform / triangulation (same pose in) and the reference-order solver are compared bit for bit;
the fast solver within the tolerances of test_gpu_parity.py.  Sizes are small: the whole file runs in seconds."""
import numpy as np
import pytest

from oracle.oracle import Camera as OCam

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 767, 769, 1023, 1025, 2047, 2049, 4097]


def _frame(vo, n, seed):
    # dropped detections, distractor points and holes in the model: ragged sets, unmatched queries, unjoined pairs
    return vo.synth.frame_pair(n, seed=seed, drop=0.1 if n > 8 else 0.0, distractors=n // 16, model_drop=0.1 if n > 8 else 0.0)


@pytest.mark.parametrize("n", SIZES)
def test_operators_at_tile_edges(vo, ctx, o32, n):
    fp = _frame(vo, n, 7000 + n)
    cam_o = OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4))
    # matcher, all three forms of the search, both argument orders (the larger set is the tree)
    exp_m = o32.match(fp["ref_app"], fp["cur_app"])
    exp_m_rev = o32.match(fp["cur_app"], fp["ref_app"])
    for mode in (1, 2, 3):
        assert ctx.lib.vo_match_set_mode(ctx.h, mode) == 0
        assert np.array_equal(vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=ctx), exp_m), mode
        assert np.array_equal(vo.compute_correspondences_images(fp["cur_app"], fp["ref_app"], ctx=ctx), exp_m_rev), mode
    assert ctx.lib.vo_match_set_mode(ctx.h, 0) == 0
    m = vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=ctx)
    assert np.array_equal(m, exp_m)
    # join
    j = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
    assert np.array_equal(j, o32.join(m, fp["model_pairs"]))
    assert np.array_equal(j, o32.join(m, fp["model_pairs"], linear=True))
    # rigid transform and projection: bit-exact
    xt = vo.transform_points(fp["X_gt"], fp["model"], ctx=ctx)
    assert np.array_equal(xt, o32.transform_points(fp["X_gt"], fp["model"]))
    cam = vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], fp["X_gt"], ctx=ctx)
    cam_gt = OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], fp["X_gt"])
    for keep in (True, False):
        uv, n_in = cam.projectPoints(fp["model"], keep_indices=keep)
        e_uv, e_in = o32.project_points(cam_gt, fp["model"], keep_indices=keep)
        assert n_in == e_in and np.array_equal(uv, e_uv)
    # triangulation with the appearances carried along: survivors, order, points bit-exact
    xyz, pairs, app = vo.triangulate_points(fp["K"], fp["X_gt"], m, fp["ref_pts"], fp["cur_pts"], fp["cur_app"], ctx=ctx)
    e_xyz, e_pairs, e_app = o32.triangulate(fp["K"], fp["X_gt"], m, fp["ref_pts"], fp["cur_pts"], fp["cur_app"])
    assert np.array_equal(pairs, e_pairs) and np.array_equal(app, e_app) and np.array_equal(xyz, e_xyz)
    # solver: reference-order arithmetic bit for bit, fast mode within tolerance; both chi^2 branches (threshold 60)
    if len(j) == 0:
        return
    for thr, keep in ((10000.0, False), (60.0, True)):
        r = o32.picp_solve_raw(cam_o, fp["model"], fp["cur_pts"], j, 6, thr, keep)
        for exact in (True, False):
            s = vo.PICPSolver(ctx)
            s.setExact(exact)
            s.setKernelThreshold(thr)
            s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
            s.solve(j, keep, 6)
            T = s.camera().worldInCameraPose()
            H, b = s.system()
            if exact:
                assert np.array_equal(T, r["T"][-1]) and np.array_equal(H, r["H"][-1]) and np.array_equal(b, r["b"][-1])
                assert s.numInliers() == int(r["stats"][-1, 2])
                assert np.float32(s.chiInliers()) == r["stats"][-1, 0] and np.float32(s.chiOutliers()) == r["stats"][-1, 1]
            elif len(j) >= 16:      # below that the normal equations are near-singular: rounding is amplified without bound
                assert np.abs(T - r["T"][-1]).max() < 1e-4 * max(1.0, float(np.abs(r["T"][-1]).max()))
            s.close()


@pytest.mark.parametrize("n,F", [(700, 1), (700, 2), (700, 7), (700, 8), (2500, 8), (2500, 15), (2500, 17), (700, 24)])
def test_batched_frames_at_frame_count_edges(vo, o32, n, F):
    """vo_frames_batch_dev maps (frame, workgroup) onto a plain 2-D grid below 8 frames and onto an XCD-aware 1-D grid from
    8 frames on, with idle padding workgroups when F is not a multiple of 8: every frame must equal the oracle's frame."""
    c = vo.Context(0)
    fps = []
    for i in range(F):
        f = vo.synth.frame_pair(n, seed=7600 + 33 * i + n, distractors=13)      # 33: same seed % 3, same set sizes
        keep = np.random.default_rng(100 + i).permutation(n)[: n - n // 8]
        f["model_pairs"] = np.ascontiguousarray(f["model_pairs"][np.sort(keep)])
        fps.append(f)
    bp = vo.BatchPipeline(c, fps, n_iters=7, kernel_threshold=10000.0)
    bp.run()
    poses, stats, counts = bp.poses(), bp.stats(), bp.counts()
    for i, f in enumerate(fps):
        m_o = o32.match(f["ref_app"], f["cur_app"]); j_o = o32.join(m_o, f["model_pairs"])
        assert np.array_equal(bp.fetch("match", i), m_o) and np.array_equal(bp.fetch("join", i), j_o), i
        assert counts[0, i] == len(m_o) and counts[1, i] == len(j_o)
        r = o32.picp_solve(OCam(480, 640, 0, 10, f["K"], np.eye(4)), f["model"], f["cur_pts"], j_o, 7, 10000.0, False, trace=False)
        assert np.abs(poses[i] - r["T"]).max() < 1e-4 and int(stats[i, 2]) == r["num_inliers"], i
        xo, po, ao = o32.triangulate(f["K"], poses[i], m_o, f["ref_pts"], f["cur_pts"], f["cur_app"])
        assert np.array_equal(bp.fetch("tri_pairs", i), po) and np.array_equal(bp.fetch("tri_app", i), ao), i
        assert np.array_equal(bp.fetch("tri_xyz", i), xo), i
    bp.close(); c.close()


@pytest.mark.parametrize("n", [300, 3000, 9000])
def test_matcher_with_non_finite_rows_and_radius_sweep(vo, ctx, o32, n):
    """Rows holding NaN or +-inf can never satisfy d2 < r^2 (brute_force_search.h:22-41 read literally) and must not
    disturb the bucketing of the finite rows, whatever the search form; the radius is a parameter, not a constant."""
    fp = vo.synth.frame_pair(n, seed=8100 + n, distractors=n // 10)
    a, b = fp["ref_app"].copy(), fp["cur_app"].copy()
    rng = np.random.default_rng(n)
    for arr in (a, b):
        rows = rng.permutation(len(arr))[:45]
        arr[rows[:15], rng.integers(0, 10, 15)] = np.nan
        arr[rows[15:30], rng.integers(0, 10, 15)] = np.inf
        arr[rows[30:40], rng.integers(0, 10, 10)] = -np.inf
        arr[rows[40:45]] = np.nan                                  # whole rows
    for radius in (0.1, 0.02, 0.45):
        exp = o32.match(a, b, radius)
        assert 0 < len(exp) < min(len(a), len(b))
        for mode in (0, 1, 2, 3):
            assert ctx.lib.vo_match_set_mode(ctx.h, mode) == 0
            got = vo.compute_correspondences_images(a, b, radius, ctx=ctx)
            assert np.array_equal(got, exp), (mode, radius, len(got), len(exp))
    assert ctx.lib.vo_match_set_mode(ctx.h, 0) == 0
    # fullSearch (all points inside the radius) on the same data
    got = vo.radius_search(a, b, 0.1, ctx=ctx)
    exp = o32.radius_search(a, b, 0.1, brute=True)
    assert len(got) == len(exp) == len(b) and all(np.array_equal(g, e) for g, e in zip(got, exp))


@pytest.mark.parametrize("seed,angle,t", [(41, 0.5, 0.8), (42, 0.3, 1.0), (43, 0.7, 0.4)])
def test_general_motions(vo, ctx, o32, seed, angle, t):
    """Rotations of tenths of a radian between the views (the other synthetic frames move by 0.05 rad): the first solver rounds
    take large steps (v2tEuler far from the identity), the triangulation sees a general relative pose, the eight-point
    initialisation a general essential matrix.  GPU against oracle and against the generating motion."""
    from oracle import vo_pipeline as vp
    fp = vo.synth.frame_pair(400, seed=seed, noise_px=0.0, max_angle=angle, max_t=t)
    m = vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=ctx)
    assert np.array_equal(m, fp["gt_matches"])
    j = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
    Xg = fp["X_gt"].astype(np.float64)
    cam_o = OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4))
    r = o32.picp_solve_raw(cam_o, fp["model"], fp["cur_pts"], j, 60, 10000.0, False)
    for exact in (True, False):
        s = vo.PICPSolver(ctx)
        s.setExact(exact)
        s.setKernelThreshold(10000.0)
        s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
        T = []
        for _ in range(60):
            s.oneRound(j, False)
            T.append(s.camera().worldInCameraPose().copy())
        T = np.array(T)
        if exact:
            assert np.array_equal(T, r["T"])                              # every round of the descent, bit for bit
        else:
            assert np.abs(T - r["T"]).max() < 2e-4 and np.abs(T[-1] - r["T"][-1]).max() < 2e-5
        assert np.abs(T[-1] - Xg).max() < 1e-4 and s.numInliers() == len(j)
        s.close()
    xyz, pairs, _ = vo.triangulate_points(fp["K"], fp["X_gt"], m, fp["ref_pts"], fp["cur_pts"], ctx=ctx)
    e_xyz, e_pairs, _ = o32.triangulate(fp["K"], fp["X_gt"], m, fp["ref_pts"], fp["cur_pts"])
    assert np.array_equal(pairs, e_pairs) and np.array_equal(xyz, e_xyz)
    X = vo.estimate_transform(fp["K"], m, fp["ref_pts"], fp["cur_pts"], ctx=ctx)
    Xo = vp.estimate_transform(o32, fp["K"], m, fp["ref_pts"], fp["cur_pts"])
    d, dg = X[:3, 3] / np.linalg.norm(X[:3, 3]), Xg[:3, 3] / np.linalg.norm(Xg[:3, 3])
    assert np.abs(X - Xo).max() < 5e-5 and np.abs(X[:3, :3] - Xg[:3, :3]).max() < 1e-4 and float(d @ dg) > 1 - 1e-7


@pytest.mark.parametrize("scale,offset,radius", [(100.0, 5000.0, 10.0), (1e-3, 0.0, 1e-4), (1.0, -3e4, 0.1), (255.0, 0.0, 20.0)])
def test_matcher_on_scaled_and_shifted_descriptors(vo, ctx, o32, scale, offset, radius):
    """Descriptors need not live in [-1, 1]: the bucketing of the pruned searches derives its cells from the data's bounds and
    the radius, and an offset of 3e4 leaves about 3 decimal digits below the radius in float32 -- the search must still return the
    oracle's pairs (distances are evaluated on the same float values on both sides)."""
    fp = vo.synth.frame_pair(4000, seed=8800, distractors=200)
    a = (fp["ref_app"].astype(np.float64) * scale + offset).astype(np.float32)
    b = (fp["cur_app"].astype(np.float64) * scale + offset).astype(np.float32)
    exp = o32.match(a, b, radius)
    assert len(exp) > 3000
    for mode in (0, 1, 2, 3):
        assert ctx.lib.vo_match_set_mode(ctx.h, mode) == 0
        got = vo.compute_correspondences_images(a, b, radius, ctx=ctx)
        assert np.array_equal(got, exp), (mode, len(got), len(exp))
    assert ctx.lib.vo_match_set_mode(ctx.h, 0) == 0


def test_non_finite_points_go_through_like_in_the_reference(vo, ctx, o32):
    """NaN / inf coordinates: the reference's gates are written as `if (x < lo || x > hi) reject`, which a NaN PASSES (every
    comparison is false) -- a NaN point is projected "inside", triangulated "in front", and carried on.  Same here, element for
    element (NaN where the oracle has NaN, equal bits elsewhere, equal counts)."""
    fp = vo.synth.frame_pair(600, seed=8900)
    rng = np.random.default_rng(3)
    world = fp["model"].copy()
    bad = rng.permutation(len(world))[:30]
    world[bad[:10], rng.integers(0, 3, 10)] = np.nan
    world[bad[10:20], rng.integers(0, 3, 10)] = np.inf
    world[bad[20:30], rng.integers(0, 3, 10)] = -np.inf
    same = lambda a, b: a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a, b, equal_nan=True)
    assert same(vo.transform_points(fp["X_gt"], world, ctx=ctx), o32.transform_points(fp["X_gt"], world))
    cam = vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], fp["X_gt"], ctx=ctx)
    cam_o = OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], fp["X_gt"])
    for keep in (True, False):
        uv, n_in = cam.projectPoints(world, keep_indices=keep)
        e_uv, e_in = o32.project_points(cam_o, world, keep_indices=keep)
        assert n_in == e_in and same(uv, e_uv), keep
    assert np.isnan(e_uv).any()                                            # a NaN point did pass the gates
    p1, p2 = fp["ref_pts"].copy(), fp["cur_pts"].copy()
    p1[bad[:10], 0] = np.nan; p2[bad[10:20], 1] = np.inf; p1[bad[20:25], 1] = -np.inf
    m = fp["gt_matches"]
    xyz, pairs, app = vo.triangulate_points(fp["K"], fp["X_gt"], m, p1, p2, fp["cur_app"], ctx=ctx)
    e_xyz, e_pairs, e_app = o32.triangulate(fp["K"], fp["X_gt"], m, p1, p2, fp["cur_app"])
    assert np.array_equal(pairs, e_pairs) and np.array_equal(app, e_app) and same(xyz, e_xyz)
    assert np.isnan(e_xyz).any()
    # the solver: a NaN landmark among the correspondences poisons H, b and the pose, in both arithmetic modes, like the oracle's loop
    j = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
    r = o32.picp_solve_raw(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)), world, fp["cur_pts"], j, 3, 10000.0, False)
    assert np.isnan(r["T"][-1]).any()
    for exact in (True, False):
        s = vo.PICPSolver(ctx)
        s.setExact(exact); s.setKernelThreshold(10000.0)
        s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), world, fp["cur_pts"])
        s.solve(j, False, 3)
        T = s.camera().worldInCameraPose()
        assert np.array_equal(np.isnan(T), np.isnan(r["T"][-1])), exact
        s.close()


def test_denormal_range_values_round_like_on_the_cpu(vo, ctx, o32):
    """float32 subnormals are neither flushed on input nor on output (the reference's SSE2 build honours them): a rigid
    transform of points of size 1e-39, and a matcher whose squared distances and squared radius are subnormal."""
    rng = np.random.default_rng(6)
    pts = (rng.uniform(-1, 1, (5000, 3)) * 1e-39).astype(np.float32)
    assert (np.abs(pts[pts != 0]) < 1.2e-38).all()
    fp = vo.synth.frame_pair(10, seed=1)
    X = fp["X_gt"].copy(); X[:3, 3] = 0
    got, exp = vo.transform_points(X, pts, ctx=ctx), o32.transform_points(X, pts)
    assert np.array_equal(got, exp) and np.count_nonzero(exp) > 14000
    base = rng.uniform(-1, 1, (3000, 10))
    a = (base * 1e-19).astype(np.float32)
    b = ((base[rng.permutation(3000)] + rng.normal(0, 1e-2, (3000, 10))) * 1e-19).astype(np.float32)
    radius = 1e-20                                                        # r^2 = 1e-40: subnormal
    exp_m = o32.match(a, b, radius)
    assert 100 < len(exp_m) <= 3000
    for mode in (0, 1, 2, 3):
        assert ctx.lib.vo_match_set_mode(ctx.h, mode) == 0
        assert np.array_equal(vo.compute_correspondences_images(a, b, radius, ctx=ctx), exp_m), mode
    assert ctx.lib.vo_match_set_mode(ctx.h, 0) == 0
