"""Invalid arguments at the C ABI: every host-pointer entry point is called with one argument broken at a time (NULL where
an array or a result is required, a negative count, a NULL handle) and must come back with a negative status and a message --
never a crash, never VO_OK.  The reference has undefined behaviour for all of these; a drop-in library must not."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_invalid_arguments_are_refused_one_by_one(vo, ctx):
    lib, h = ctx.lib, ctx.h
    fp = vo.synth.frame_pair(300, seed=9300)
    K = np.ascontiguousarray(fp["K"].astype(np.float32).T); T = np.ascontiguousarray(np.eye(4, dtype=np.float32))
    world = np.ascontiguousarray(fp["model"], np.float32); uv = np.ascontiguousarray(fp["cur_pts"], np.float32)
    a1 = np.ascontiguousarray(fp["ref_app"], np.float32); a2 = np.ascontiguousarray(fp["cur_app"], np.float32)
    p1 = np.ascontiguousarray(fp["ref_pts"], np.float32)
    pairs = np.ascontiguousarray(fp["gt_matches"], np.int32); mp = np.ascontiguousarray(fp["model_pairs"], np.int32)
    n = len(world)
    out2 = np.zeros((n, 2), np.float32); out3 = np.zeros((n, 3), np.float32); outp = np.zeros((n, 2), np.int32); outa = np.zeros((n, 10), np.float32)
    offs = np.zeros(n + 1, np.int32); idx = np.zeros(8 * n, np.int32); X = np.zeros(16, np.float32)
    n_out, n_in = C.c_int(), C.c_int()
    I, F, NUL = C.c_int, C.c_float, C.c_void_p(0)
    calls = {
        # name: (argument list of a VALID call, indices that must not be NULL, indices of counts that must not be negative)
        "vo_project_points": ([h, I(480), I(640), I(0), I(10), _p(K), _p(T), _p(world), I(n), I(1), _p(out2), C.byref(n_out), C.byref(n_in)],
                              [0, 5, 6, 7, 10], [8]),
        "vo_match_appearances": ([h, _p(a1), I(len(a1)), _p(a2), I(len(a2)), F(0.1), _p(outp), C.byref(n_out)], [0, 1, 3, 6, 7], [2, 4]),
        "vo_radius_search": ([h, _p(a1), I(len(a1)), _p(a2), I(len(a2)), F(0.1), _p(offs), _p(idx), I(len(idx)), C.byref(n_out)],
                             [0, 1, 3, 6, 7, 9], [2, 4, 8]),
        "vo_join_correspondences": ([h, _p(pairs), I(len(pairs)), _p(mp), I(len(mp)), _p(outp), C.byref(n_out)], [0, 1, 3, 5, 6], [2, 4]),
        "vo_transform_points": ([h, _p(T), _p(world), I(n), _p(out3)], [0, 1, 2, 4], [3]),
        "vo_triangulate": ([h, _p(K), _p(T), _p(pairs), I(len(pairs)), _p(p1), I(len(p1)), _p(uv), I(len(uv)), _p(a2), _p(out3), _p(outp), _p(outa),
                            C.byref(n_out)], [0, 1, 2, 3, 5, 7, 10, 13], [4, 6, 8]),
        "vo_estimate_transform": ([h, _p(K), _p(pairs), I(len(pairs)), _p(p1), I(len(p1)), _p(uv), I(len(uv)), _p(X)], [0, 1, 2, 4, 6, 8], [3, 5, 7]),
    }
    checked = 0
    for name, (args, ptrs, counts) in calls.items():
        f = getattr(lib, name)
        f.restype = C.c_int
        assert f(*args) == 0, (name, lib.vo_last_error())                  # the baseline call is valid
        for i in ptrs:
            bad = list(args); bad[i] = NUL
            rc = f(*bad)
            assert rc < 0 and lib.vo_last_error(), (name, "NULL argument", i, rc)
            checked += 1
        for i in counts:
            bad = list(args); bad[i] = I(-1)
            rc = f(*bad)
            assert rc < 0 and lib.vo_last_error(), (name, "negative count", i, rc)
            checked += 1
        assert f(*args) == 0, (name, "valid call after the refused ones", lib.vo_last_error())
    assert checked == 54

    # the solver handle
    s = C.c_void_p()
    assert lib.vo_picp_create(h, C.byref(s)) == 0
    assert lib.vo_picp_create(NUL, C.byref(s)) < 0 and lib.vo_picp_create(h, NUL) < 0
    assert lib.vo_picp_one_round(s, _p(pairs), I(len(pairs)), I(0)) < 0                     # before init: not ready
    assert lib.vo_picp_set_camera(s, I(480), I(640), I(0), I(10), NUL, _p(T)) < 0
    assert lib.vo_picp_set_camera(s, I(480), I(640), I(0), I(10), _p(K), NUL) < 0
    assert lib.vo_picp_set_camera(s, I(480), I(640), I(0), I(10), _p(K), _p(T)) == 0
    assert lib.vo_picp_set_points(s, NUL, I(n), _p(uv), I(len(uv))) < 0
    assert lib.vo_picp_set_points(s, _p(world), I(-1), _p(uv), I(len(uv))) < 0
    assert lib.vo_picp_set_points(s, _p(world), I(n), NUL, I(len(uv))) < 0
    assert lib.vo_picp_set_points(s, _p(world), I(n), _p(uv), I(len(uv))) == 0
    j = np.ascontiguousarray(np.stack([np.arange(50), np.arange(50)], 1), np.int32)
    assert lib.vo_picp_one_round(s, NUL, I(50), I(0)) < 0
    assert lib.vo_picp_one_round(s, _p(j), I(-1), I(0)) < 0
    assert lib.vo_picp_solve(s, _p(j), I(50), I(0), I(-1)) < 0
    assert lib.vo_picp_one_round(NUL, _p(j), I(50), I(0)) < 0
    assert lib.vo_picp_one_round(s, _p(j), I(50), I(0)) == 0
    assert lib.vo_picp_get_pose(s, NUL) < 0 and lib.vo_picp_get_pose(NUL, _p(X)) < 0
    assert lib.vo_picp_get_stats(s, NUL, NUL, NUL) <= 0                                     # all-NULL outputs: refused or a no-op, not a crash
    assert lib.vo_picp_get_pose(s, _p(X)) == 0 and np.isfinite(X).all()
    assert lib.vo_picp_destroy(s) == 0
    assert lib.vo_picp_destroy(NUL) <= 0
    # handles of the wrong kind are NULL-checked only; contexts
    assert lib.vo_ctx_synchronize(NUL) < 0 and lib.vo_ctx_destroy(NUL) <= 0 and lib.vo_match_set_mode(NUL, I(1)) < 0
    assert lib.vo_match_set_mode(h, I(9)) < 0 and lib.vo_picp_batch_set_form(h, I(7)) < 0
    kd = C.c_void_p()
    assert lib.vo_kdtree_create(h, NUL, I(10), I(20), C.byref(kd)) < 0
    assert lib.vo_kdtree_create(h, _p(a1), I(-1), I(20), C.byref(kd)) < 0
    assert lib.vo_kdtree_create(h, _p(a1), I(len(a1)), I(20), NUL) < 0
    assert lib.vo_kdtree_create(h, _p(a1), I(len(a1)), I(20), C.byref(kd)) == 0
    best = np.zeros(len(a2), np.int32)
    assert lib.vo_kdtree_best_match_fast(kd, NUL, I(len(a2)), F(0.1), _p(best)) < 0
    assert lib.vo_kdtree_best_match_fast(kd, _p(a2), I(-1), F(0.1), _p(best)) < 0
    assert lib.vo_kdtree_best_match_fast(kd, _p(a2), I(len(a2)), F(0.1), NUL) < 0
    assert lib.vo_kdtree_best_match_fast(NUL, _p(a2), I(len(a2)), F(0.1), _p(best)) < 0
    assert lib.vo_kdtree_best_match_fast(kd, _p(a2), I(len(a2)), F(0.1), _p(best)) == 0
    assert lib.vo_kdtree_destroy(kd) == 0


def test_misuse_of_capture_allocation_and_batches(vo, o32):
    """API misuse that must be refused and leave the context usable: capture begun twice / ended without a begin / a host
    entry point inside a capture, an allocation far beyond the card, a frame batch with missing arrays or nonsense sizes."""
    c = vo.Context(0)
    lib, h = c.lib, c.h
    NUL, I = C.c_void_p(0), C.c_int
    g = C.c_void_p()
    assert lib.vo_ctx_end_capture(h, C.byref(g)) < 0                       # nothing to end
    assert lib.vo_ctx_begin_capture(h) == 0
    assert lib.vo_ctx_begin_capture(h) < 0                                 # already capturing
    fp = vo.synth.frame_pair(200, seed=9400)
    a1 = np.ascontiguousarray(fp["ref_app"], np.float32); a2 = np.ascontiguousarray(fp["cur_app"], np.float32)
    outp = np.zeros((200, 2), np.int32); n_out = C.c_int()
    rc = lib.vo_match_appearances(h, _p(a1), I(200), _p(a2), I(200), C.c_float(0.1), _p(outp), C.byref(n_out))
    assert rc < 0                                                          # host copies / synchronisation are not capturable
    assert lib.vo_ctx_end_capture(h, C.byref(g)) in (0, -3)                # the capture ends (possibly invalidated by the refused call)
    if g.value:
        assert lib.vo_graph_destroy(g) == 0
    assert lib.vo_ctx_end_capture(h, NUL) < 0 and lib.vo_graph_launch(NUL) < 0
    m = vo.compute_correspondences_images(a1, a2, ctx=c)                   # and the context still works
    assert np.array_equal(m, o32.match(a1, a2))
    # allocation beyond the card
    d = C.c_void_p()
    rc = lib.vo_dev_alloc(h, C.c_size_t(1 << 42), C.byref(d))              # 4 TiB
    assert rc == -4 and not d.value, rc                                    # VO_ERR_OUT_OF_MEMORY
    assert lib.vo_dev_alloc(h, C.c_size_t(1 << 20), C.byref(d)) == 0 and d.value
    assert lib.vo_dev_free(h, d) == 0
    assert lib.vo_dev_alloc(NUL, C.c_size_t(16), C.byref(d)) < 0 and lib.vo_dev_alloc(h, C.c_size_t(16), NUL) < 0
    assert np.array_equal(vo.compute_correspondences_images(a1, a2, ctx=c), m)
    # frame batch: NULL batch, and a valid batch broken one field at a time
    assert lib.vo_frames_batch_dev(h, NUL) < 0
    fps = [vo.synth.frame_pair(300, seed=9400 + 3 * k) for k in range(2)]
    bp = vo.BatchPipeline(c, fps, n_iters=3)
    bp.run()
    ok_poses = bp.poses().copy()
    b = bp.b
    for field, bad in (("n_frames", -1), ("n_ref", -1), ("n_cur", -5), ("n_model", -1), ("n_model_pairs", -2), ("n_iters", -1),
                       ("ref_app", None), ("cur_pts", None), ("model", None), ("model_pairs", None), ("matches", None), ("joined", None),
                       ("poses", None), ("tri_xyz", None), ("tri_pairs", None), ("counts", None)):
        keep = getattr(b, field)
        setattr(b, field, bad)
        rc = lib.vo_frames_batch_dev(h, C.byref(b))
        setattr(b, field, keep)
        assert rc < 0, (field, rc)
    bp.run()
    assert np.array_equal(bp.poses(), ok_poses)
    bp.close(); c.close()


def test_wild_indices_are_dropped_not_dereferenced(vo, ctx, o32):
    """Index pairs pointing outside their arrays (negative, INT_MAX, INT_MIN, one past the end): undefined behaviour in the
    reference, a dropped pair here -- the result equals the oracle's on the input with those pairs removed."""
    fp = vo.synth.frame_pair(1200, seed=9500, drop=0.05, model_drop=0.05)
    rng = np.random.default_rng(4)
    m = o32.match(fp["ref_app"], fp["cur_app"])
    wild = np.array([-1, -7, 2**31 - 1, -2**31, len(fp["ref_pts"]), len(fp["cur_pts"]) + 5], np.int64)

    def spoil(pairs, n_first, n_second):
        p = pairs.astype(np.int64).copy()
        rows = rng.permutation(len(p))[:60]
        p[rows[:30], 0] = rng.choice(wild, 30); p[rows[30:], 1] = rng.choice(wild, 30)
        p = p.astype(np.int32)
        good = (p[:, 0] >= 0) & (p[:, 0] < n_first) & (p[:, 1] >= 0) & (p[:, 1] < n_second)
        return np.ascontiguousarray(p), np.ascontiguousarray(p[good])

    # triangulation: (index in p1, index in p2)
    bad, clean = spoil(m, len(fp["ref_pts"]), len(fp["cur_pts"]))
    assert len(clean) < len(bad)
    xyz, pairs, app = vo.triangulate_points(fp["K"], fp["X_gt"], bad, fp["ref_pts"], fp["cur_pts"], fp["cur_app"], ctx=ctx)
    e_xyz, e_pairs, e_app = o32.triangulate(fp["K"], fp["X_gt"], clean, fp["ref_pts"], fp["cur_pts"], fp["cur_app"])
    assert np.array_equal(pairs, e_pairs) and np.array_equal(xyz, e_xyz) and np.array_equal(app, e_app)
    # join: image pairs (ref, cur) with world pairs (ref, model); a wild ref on either side finds no partner
    mp = fp["model_pairs"]
    bad_img, _ = spoil(m, 1 << 30, 1 << 30)                               # only the sign / size of .first matters to the join
    bad_img[:, 1] = m[:, 1]
    n_ref = len(fp["ref_pts"])
    ok_img = bad_img[(bad_img[:, 0] >= 0) & (bad_img[:, 0] < n_ref)]          # the oracle would index its table with the wild ones
    bad_mp = mp.astype(np.int64).copy(); rows = rng.permutation(len(mp))[:20]; bad_mp[rows, 0] = rng.choice(wild[:4], 20); bad_mp = bad_mp.astype(np.int32)
    ok_mp = bad_mp[(bad_mp[:, 0] >= 0) & (bad_mp[:, 0] < n_ref)]
    j = vo.extract_correspondences_world(bad_img, bad_mp, ctx=ctx)
    assert np.array_equal(j, o32.join(ok_img, ok_mp, linear=True))
    # solver: reported, not dereferenced
    s = vo.PICPSolver(ctx)
    s.setKernelThreshold(10000.0)
    s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
    jj = o32.join(m, mp)
    spoiled = jj.copy(); spoiled[5, 1] = 2**31 - 1; spoiled[9, 0] = -3
    s.oneRound(spoiled, False)
    with pytest.raises(vo.VoError) as e:
        s.numInliers()
    assert e.value.code == -5                                              # VO_ERR_BAD_INDEX
    s.close()


def test_handles_that_outlive_their_context(vo, o32):
    """A context destroyed BEFORE the solver / kd-tree / graph made on it (static destruction order, a script closing things
    in the wrong order): the survivors must fail cleanly when used and must still be destroyable -- no use-after-free."""
    c = vo.Context(0)
    lib = c.lib
    fp = vo.synth.frame_pair(400, seed=9600)
    s = vo.PICPSolver(c)
    s.setKernelThreshold(10000.0)
    s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=c), fp["model"], fp["cur_pts"])
    j = o32.join(o32.match(fp["ref_app"], fp["cur_app"]), fp["model_pairs"])
    s.solve(j, False, 5)
    T = s.camera().worldInCameraPose().copy()
    kd = vo.KdTree(fp["ref_app"], ctx=c)
    p = vo.FramePipeline(c, fp, n_iters=3)
    p.frame(); p.capture_frame(); p.frame_graph(); p.counts()
    graph = p.graph if hasattr(p, "graph") else None
    h_ctx = c.h
    assert lib.vo_ctx_destroy(h_ctx) == 0
    c.h = None                                               # the Python object must not destroy it again
    assert lib.vo_ctx_destroy(h_ctx) < 0                     # destroyed twice: refused, not a double free
    with pytest.raises(vo.VoError):
        s.oneRound(j, False)
    with pytest.raises(vo.VoError):
        s.camera()
    with pytest.raises(vo.VoError):
        kd.bestMatchFast(fp["cur_app"])
    if graph is not None:
        assert lib.vo_graph_launch(graph) < 0
    s.close(); kd.close()                                     # still destroyable
    c2 = vo.Context(0)                                        # and the library goes on
    p.ctx = c2                                                # the pipeline's graph and solver go the same way, its buffers through a live context
    p.close()
    s2 = vo.PICPSolver(c2)
    s2.setKernelThreshold(10000.0)
    s2.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=c2), fp["model"], fp["cur_pts"])
    s2.solve(j, False, 5)
    assert np.array_equal(s2.camera().worldInCameraPose(), T)
    s2.close(); c2.close()


def test_map_and_device_initialisation_refuse_bad_arguments(vo, o32):
    """round 5's entry points: nulls, negative counts, misaligned device rows, a map that outlives its context, a map
    asked to grow inside a graph capture, fewer than eight pairs / wild pair indices for vo_estimate_transform_dev"""
    c = vo.Context(0)
    lib = c.lib
    I, NUL = C.c_int, C.c_void_p(0)
    h = C.c_void_p()
    assert lib.vo_map_create(NUL, I(0), C.byref(h)) < 0 and lib.vo_map_create(c.h, I(-1), C.byref(h)) < 0
    assert lib.vo_map_create(c.h, I((1 << 29) + 1), C.byref(h)) < 0 and lib.vo_map_create(c.h, I(0), NUL) < 0
    m = vo.Map(c, capacity=1024)
    rng = np.random.default_rng(3)
    pts = rng.normal(0, 1, (600, 3)).astype(np.float32); app = rng.uniform(-1, 1, (600, 10)).astype(np.float32)
    d_p, d_a = c.to_device(pts), c.to_device(app)
    assert lib.vo_map_update_dev(NUL, C.c_void_p(d_p), C.c_void_p(d_a), I(600), NUL, NUL) < 0
    assert lib.vo_map_update_dev(m.h, NUL, C.c_void_p(d_a), I(600), NUL, NUL) < 0
    assert lib.vo_map_update_dev(m.h, C.c_void_p(d_p), NUL, I(600), NUL, NUL) < 0
    assert lib.vo_map_update_dev(m.h, C.c_void_p(d_p), C.c_void_p(d_a), I(-5), NUL, NUL) < 0
    assert lib.vo_map_update_dev(m.h, C.c_void_p(d_p), C.c_void_p(d_a + 4), I(599), NUL, NUL) < 0        # rows off the 8-byte boundary
    assert lib.vo_map_update(m.h, NUL, _p(app), I(600), NUL) < 0 and lib.vo_map_update(m.h, _p(pts), _p(app), I(-1), NUL) < 0
    assert lib.vo_map_history_reset_dev(m.h, NUL) < 0 and lib.vo_map_history_step_dev(m.h, NUL) < 0 and lib.vo_map_size(m.h, NUL) < 0
    assert lib.vo_map_update_dev(m.h, C.c_void_p(d_p), C.c_void_p(d_a), I(0), NUL, NUL) == 0 and len(m) == 0   # an empty cloud is fine
    m.update(pts, app)
    assert len(m) == 600
    # inside a graph capture the map may be updated while it has room, and refuses to grow
    assert lib.vo_ctx_begin_capture(c.h) == 0
    try:
        assert lib.vo_map_update_dev(m.h, C.c_void_p(d_p), C.c_void_p(d_a), I(400), NUL, NUL) == 0      # 600 + 400 <= 1024 by the host's bound
        assert lib.vo_map_update_dev(m.h, C.c_void_p(d_p), C.c_void_p(d_a), I(600), NUL, NUL) < 0       # would have to ask the device and grow
        assert lib.vo_map_size(m.h, C.byref(I())) < 0                                                   # waits: not inside a capture
    finally:
        g = C.c_void_p()
        assert lib.vo_ctx_end_capture(c.h, C.byref(g)) == 0
    assert len(m) == 600                                     # a capture records, it does not run
    assert lib.vo_graph_launch(g) == 0 and len(m) == 600     # the recorded update brought the first 400 rows again: all known
    lib.vo_graph_destroy(g)
    # vo_estimate_transform_dev
    seq = vo.synth.sequence(seed=5, n_frames=2, n_visible=200)
    f0, f1 = seq["frames"]
    corr = o32.match(f0["app"], f1["app"])
    K = np.ascontiguousarray(np.asarray(seq["K"], np.float32).T); X = np.zeros(16, np.float32)
    d_c, d_0, d_1 = c.to_device(corr), c.to_device(f0["pts"]), c.to_device(f1["pts"])
    ok = [c.h, _p(K), C.c_void_p(d_c), I(len(corr)), NUL, C.c_void_p(d_0), I(len(f0["pts"])), C.c_void_p(d_1), I(len(f1["pts"])), _p(X)]
    assert lib.vo_estimate_transform_dev(*ok) == 0 and np.isfinite(X).all()
    for k in (0, 1, 2, 5, 7, 9):
        a = list(ok); a[k] = NUL
        assert lib.vo_estimate_transform_dev(*a) < 0, k
    for k, v in ((3, 7), (3, -1), (6, 0), (8, -3)):
        a = list(ok); a[k] = I(v)
        assert lib.vo_estimate_transform_dev(*a) == -1, (k, v)
    d_few = c.to_device(np.array([5], np.int32))            # the device-side count says five pairs: refused after the read-back
    a = list(ok); a[4] = C.c_void_p(d_few)
    assert lib.vo_estimate_transform_dev(*a) == -1
    bad = corr.copy(); bad[3, 1] = 10 ** 6
    c.h2d(d_c, bad)
    assert lib.vo_estimate_transform_dev(*ok) == -5 and b"outside" in lib.vo_last_error()
    # a map that outlives its context fails cleanly and can still be destroyed
    h_ctx = c.h
    for d in (d_p, d_a, d_c, d_0, d_1, d_few):
        c.free(d)
    assert lib.vo_ctx_destroy(h_ctx) == 0
    c.h = None
    with pytest.raises(vo.VoError):
        m.update(pts, app)
    with pytest.raises(vo.VoError):
        len(m)
    m.close()
