"""Further GPU parity cases: general (non-pinhole) K, many correspondences per
thread (grid-stride path), a solver reused across sizes, keep_outliers in the
batched solver, empty problems in a batch, the device-resident frame pipeline."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import rel_err
from oracle.oracle import Camera as OCam

pytestmark = pytest.mark.gpu


def _corr(fp):
    mp = np.full(len(fp["ref_app"]), -1, np.int64)
    mp[fp["model_pairs"][:, 0]] = fp["model_pairs"][:, 1]
    gt = fp["gt_matches"]
    ok = mp[gt[:, 0]] >= 0
    return np.stack([gt[ok, 1], mp[gt[ok, 0]]], axis=1).astype(np.int32)


def test_general_camera_matrix(vo, ctx, o32, o64):
    """K with skew and a non-unit K22: the general 3x3 instantiation (not the pinhole one)."""
    fp = vo.synth.frame_pair(3000, seed=301, drop=0.05, distractors=5, model_drop=0.05)
    K = fp["K"].copy()
    K[0, 1] = 0.7            # skew
    K[2, 2] = 1.0009765625   # exactly representable, != 1
    corr = _corr(fp)
    for thr, keep in ((10000.0, False), (40.0, True)):
        cam = vo.Camera(480, 640, 0, 10, K, np.eye(4), ctx=ctx)
        s = vo.PICPSolver(ctx); s.setKernelThreshold(thr)
        s.init(cam, fp["model"], fp["cur_pts"])
        s.oneRound(corr, keep)
        H, b = s.system()
        ocam = OCam(480, 640, 0, 10, K, np.eye(4))
        r64 = o64.picp_solve(ocam, fp["model"], fp["cur_pts"], corr, 6, thr, keep)
        r32 = o32.picp_solve(ocam, fp["model"], fp["cur_pts"], corr, 6, thr, keep)
        assert rel_err(H - np.eye(6, dtype=np.float32), r64["H"][0]) < 1e-5 and rel_err(b, r64["b"][0]) < 1e-5
        assert s.numInliers() == int(r32["stats"][0, 2])
        s.solve(corr, keep, 5)
        assert np.abs(s.camera().worldInCameraPose() - r32["T"]).max() < 1e-4
        assert s.numInliers() == r32["num_inliers"]
        s.close()


def test_many_correspondences_per_thread_and_reuse(vo, ctx, o32):
    """300k correspondences: the grid is capped at 4 workgroups per CU, threads loop; then the same
    solver handle is reused for a much smaller problem (partial buffers must be re-zeroed)."""
    fp = vo.synth.frame_pair(20000, seed=302)
    base = _corr(fp)
    big = np.concatenate([base] * 15)                     # repeats are legal: 300k terms
    cam = vo.Camera(480, 640, 0, 10, fp["K"], np.eye(4), ctx=ctx)
    s = vo.PICPSolver(ctx); s.setKernelThreshold(10000.0)
    s.init(cam, fp["model"], fp["cur_pts"])
    s.solve(big, False, 3)
    ocam = OCam(480, 640, 0, 10, fp["K"], np.eye(4))
    r = o32.picp_solve(ocam, fp["model"], fp["cur_pts"], big, 3, 10000.0, False, trace=False)
    assert s.numInliers() == r["num_inliers"] == len(big)
    assert np.abs(s.camera().worldInCameraPose() - r["T"]).max() < 1e-4
    small = base[:300]
    cam2 = vo.Camera(480, 640, 0, 10, fp["K"], np.eye(4), ctx=ctx)
    s.init(cam2, fp["model"], fp["cur_pts"])
    s.solve(small, False, 4)
    r2 = o32.picp_solve(ocam, fp["model"], fp["cur_pts"], small, 4, 10000.0, False, trace=False)
    assert s.numInliers() == r2["num_inliers"] == 300
    assert np.abs(s.camera().worldInCameraPose() - r2["T"]).max() < 1e-4
    s.close()


@pytest.mark.parametrize("form,iters", [(1, 6), (2, 6), (0, 6), (3, 6), (1, 0), (2, 0), (3, 0), (1, 1)])
def test_batched_keep_outliers_and_empty_problems(vo, ctx, o32, form, iters):
    """every form of the batched solver (1: one launch per round, problem = grid dimension; 2: one workgroup per
    problem; 3: reference-order arithmetic), with outliers kept, an empty problem, per-problem starting poses and
    zero / one round"""
    assert ctx.lib.vo_picp_batch_set_form(ctx.h, 4) != 0 and ctx.lib.vo_picp_batch_set_form(ctx.h, form) == 0
    P, n, thr = 4, 2000, 40.0
    fps = [vo.synth.frame_pair(n, seed=5000 + p) for p in range(P)]
    pairs = [_corr(f) for f in fps]
    pairs[2] = pairs[2][:0]                               # an empty problem: pose must stay at T0
    lib = ctx.lib
    world = np.stack([f["model"] for f in fps]); meas = np.stack([f["cur_pts"] for f in fps])
    pbuf = np.zeros((P, n, 2), np.int32)
    for i, p in enumerate(pairs):
        pbuf[i, : len(p)] = p
    npairs = np.array([len(p) for p in pairs], np.int32)
    rng = np.random.default_rng(1)
    T0 = np.stack([vo.synth.random_isometry(rng, 0.01, 0.02) for _ in range(P)])
    T0_cm = np.ascontiguousarray(np.transpose(T0, (0, 2, 1))).reshape(P, 16)
    d = [ctx.to_device(a) for a in (world, meas, pbuf, npairs, T0_cm)]
    d_T = ctx.alloc(P * 64); d_stats = ctx.alloc(P * 16)
    K = np.ascontiguousarray(fps[0]["K"].T).ravel()
    rc = lib.vo_picp_solve_batch_dev(ctx.h, P, 480, 640, 0, 10, K.ctypes.data_as(C.c_void_p), C.c_float(thr), 1,
                                     C.c_void_p(d[0]), C.c_size_t(n), C.c_void_p(d[1]), C.c_size_t(n),
                                     C.c_void_p(d[2]), C.c_size_t(n), C.c_void_p(d[3]), C.c_void_p(d[4]), iters,
                                     C.c_void_p(d_T), C.c_void_p(d_stats))
    assert rc == 0, lib.vo_last_error()
    T = np.zeros((P, 16), np.float32); st = np.zeros((P, 4), np.float32)
    ctx.d2h(T, d_T); ctx.d2h(st, d_stats)
    for p in range(P):
        Tp = T[p].reshape(4, 4).T
        r = o32.picp_solve(OCam(480, 640, 0, 10, fps[p]["K"], T0[p]), fps[p]["model"], fps[p]["cur_pts"], pairs[p],
                           iters, thr, True, trace=False)
        assert np.abs(Tp - r["T"]).max() < 1e-4
        assert int(st[p, 2]) == r["num_inliers"]
        assert abs(st[p, 1] - r["chi_outliers"]) <= 1e-4 * max(1.0, r["chi_outliers"])
        if form == 3:                                     # reference-order arithmetic: bit for bit
            assert np.array_equal(Tp, r["T"]) and (iters == 0 or st[p, 1] == np.float32(r["chi_outliers"]))
    assert np.array_equal(T[2].reshape(4, 4).T, T0[2])    # H = I, b = 0: dx = 0 exactly
    for x in d + [d_T, d_stats]:
        ctx.free(x)
    ctx.lib.vo_picp_batch_set_form(ctx.h, 0)


def test_frame_pipeline_device_resident(vo, ctx, o32):
    """FramePipeline (every stage through the *_dev entry points, counts in device memory)."""
    fp = vo.synth.frame_pair(4000, seed=303, drop=0.1, distractors=50, model_drop=0.1)
    pipe = vo.FramePipeline(ctx, fp, n_iters=10, kernel_threshold=10000.0)
    for _ in range(2):                                    # run twice: state must not leak between frames
        pipe.frame()
    m, j = pipe.fetch("match"), pipe.fetch("join")
    m_o = o32.match(fp["ref_app"], fp["cur_app"]); j_o = o32.join(m_o, fp["model_pairs"])
    assert np.array_equal(m, m_o) and np.array_equal(j, j_o)
    r = o32.picp_solve(OCam(480, 640, 0, 10, fp["K"], np.eye(4)), fp["model"], fp["cur_pts"], j_o, 10, 10000.0, False,
                       trace=False)
    T = pipe.pose()
    assert np.abs(T - r["T"]).max() < 1e-4 and pipe.stats()[2] == r["num_inliers"]
    xyz, pairs, app = pipe.fetch("tri_xyz"), pipe.fetch("tri_pairs"), pipe.fetch("tri_app")
    xo, po, ao = o32.triangulate(fp["K"], T, m_o, fp["ref_pts"], fp["cur_pts"], fp["cur_app"])
    assert np.array_equal(pairs, po) and np.array_equal(app, ao)
    assert np.array_equal(xyz, xo)                          # same pose in, same operations: bit for bit
    pipe.close()


def test_whole_frame_graph_replay(vo, ctx, o32):
    """vo_ctx_begin_capture/end_capture: a captured frame replays to the same bits as plain launches,
    and follows new data in the same device buffers (counts live in device memory)."""
    fp = vo.synth.frame_pair(3000, seed=304, drop=0.1, distractors=40, model_drop=0.1)
    c2 = vo.Context(0)
    pipe = vo.FramePipeline(c2, fp, n_iters=8, kernel_threshold=10000.0)
    pipe.frame()
    ref = (pipe.pose().tobytes(), pipe.fetch("match").tobytes(), pipe.fetch("join").tobytes(), pipe.fetch("tri_xyz").tobytes())
    pipe.capture_frame()
    for _ in range(3):
        pipe.frame_graph()
    got = (pipe.pose().tobytes(), pipe.fetch("match").tobytes(), pipe.fetch("join").tobytes(), pipe.fetch("tri_xyz").tobytes())
    assert got == ref
    # new measurements in the same buffers: drop the appearance of 100 current points -> fewer matches
    cur_app = fp["cur_app"].copy()
    cur_app[:100] = 9.0
    c2.h2d(pipe.d_cur_app, cur_app)
    pipe.frame_graph()
    m_o = o32.match(fp["ref_app"], cur_app)
    assert np.array_equal(pipe.fetch("match"), m_o) and len(m_o) < len(np.frombuffer(ref[1], np.int32)) // 2
    j_o = o32.join(m_o, fp["model_pairs"])
    r = o32.picp_solve(OCam(480, 640, 0, 10, fp["K"], np.eye(4)), fp["model"], fp["cur_pts"], j_o, 8, 10000.0, False, trace=False)
    assert np.abs(pipe.pose() - r["T"]).max() < 1e-4 and pipe.stats()[2] == r["num_inliers"]
    pipe.close(); c2.close()


@pytest.mark.parametrize("n,F", [(700, 5), (6000, 3), (2500, 9)])   # full scan / bucket-pruned / cell-hash (F >= 8)
def test_batched_frames_equal_single_frames(vo, o32, n, F):
    """vo_frames_batch_dev: every frame of the batch gets exactly the matches / joined pairs / survivors of the
    per-frame path and of the oracle, and its pose within the reduction-order tolerance."""
    c = vo.Context(0)
    fps = []
    for i in range(F):                                   # same sizes, different content
        f = vo.synth.frame_pair(n, seed=7000 + 33 * i + n, distractors=(17 if F >= 8 else 0))   # 33: same seed % 3, same sizes
        keep = np.random.default_rng(i).permutation(n)[: n - n // 10]
        f["model_pairs"] = np.ascontiguousarray(f["model_pairs"][np.sort(keep)])     # some points without model
        fps.append(f)
    bp = vo.BatchPipeline(c, fps, n_iters=9, kernel_threshold=10000.0)
    for _ in range(2):
        bp.run()
    poses, stats, counts = bp.poses(), bp.stats(), bp.counts()
    for i, f in enumerate(fps):
        m_o = o32.match(f["ref_app"], f["cur_app"]); j_o = o32.join(m_o, f["model_pairs"])
        assert np.array_equal(bp.fetch("match", i), m_o) and np.array_equal(bp.fetch("join", i), j_o)
        assert counts[0, i] == len(m_o) and counts[1, i] == len(j_o)
        r = o32.picp_solve(OCam(480, 640, 0, 10, f["K"], np.eye(4)), f["model"], f["cur_pts"], j_o, 9, 10000.0, False, trace=False)
        assert np.abs(poses[i] - r["T"]).max() < 1e-4 and int(stats[i, 2]) == r["num_inliers"]
        xo, po, ao = o32.triangulate(f["K"], poses[i], m_o, f["ref_pts"], f["cur_pts"], f["cur_app"])
        assert np.array_equal(bp.fetch("tri_pairs", i), po) and np.array_equal(bp.fetch("tri_app", i), ao)
        assert np.array_equal(bp.fetch("tri_xyz", i), xo)    # same pose in: bit for bit
    bp.close(); c.close()


def test_moved_cloud_is_an_optional_output(vo, ctx, o32):
    """vo_frame_batch.model_moved == NULL: the solver's gather applies X_prev itself (PointCloud.h:80, the transform kernel's
    arithmetic) -- every result bit for bit what the call with the moved cloud gives, and the moved cloud, where asked
    for, is X_prev * model"""
    fps = [vo.synth.frame_pair(1500, seed=9100 + k) for k in range(9)]
    rng = np.random.default_rng(4)
    Xs, moved = [], []
    for f in fps:                                   # model given in another frame: model = X^-1 * (model), X_prev = X
        ang = rng.uniform(-0.3, 0.3, 3); t = rng.uniform(-0.5, 0.5, 3)
        X = np.eye(4, dtype=np.float32)
        X[:3, :3] = o32.v2t_euler(np.concatenate([np.zeros(3), ang]).astype(np.float32))[:3, :3]
        X[:3, 3] = t
        Xi = np.linalg.inv(X.astype(np.float64)).astype(np.float32)
        f["model"] = o32.transform_points(Xi, f["model"])
        Xs.append(X); moved.append(o32.transform_points(X, f["model"]))
    res = {}
    for with_moved in (True, False):
        bp = vo.BatchPipeline(ctx, fps, n_iters=12, with_moved=with_moved, X_prev=Xs)
        bp.run()
        res[with_moved] = (bp.poses().copy(), bp.stats().copy(), bp.counts().copy(), [bp.fetch("tri_xyz", f) for f in range(len(fps))])
        if with_moved:
            got = np.zeros((len(fps), len(fps[0]["model"]), 3), np.float32)
            ctx.d2h(got, bp.d_moved)
            for f in range(len(fps)):
                assert np.array_equal(got[f], moved[f]), f
        bp.close()
    a, b = res[True], res[False]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
    for f, T in enumerate(a[0]):
        assert np.abs(T - fps[f]["X_gt"]).max() < 5e-3, f      # (the moved model is the model the pair was generated with, to rounding)


def test_device_pose_reset_and_in_place_pose(vo, ctx, o32):
    """vo_picp_set_pose_dev takes effect at the next solve (folded into the gather launch, or a launch of
    its own when the correspondences are cached); vo_picp_pose_dev_ptr exposes the result in place."""
    fp = vo.synth.frame_pair(2500, seed=305)
    corr = _corr(fp)
    cam = vo.Camera(480, 640, 0, 10, fp["K"], np.eye(4), ctx=ctx)
    s = vo.PICPSolver(ctx); s.setKernelThreshold(10000.0)
    s.init(cam, fp["model"], fp["cur_pts"])
    s.solve(corr, False, 7)
    T1 = s.camera().worldInCameraPose().copy()
    d_I = ctx.to_device(np.eye(4, dtype=np.float32))
    lib = ctx.lib
    for expect_repack in (False, True):
        assert lib.vo_picp_set_pose_dev(s.h, C.c_void_p(d_I)) == 0
        if expect_repack:
            corr = corr.copy()                       # new host array: uploaded and gathered again
        s.solve(corr, False, 7)
        assert s.camera().worldInCameraPose().tobytes() == T1.tobytes()      # same start, same bits
    p = C.c_void_p()
    assert lib.vo_picp_pose_dev_ptr(s.h, C.byref(p)) == 0 and p.value
    T = np.zeros(16, np.float32)
    ctx.d2h(T, p.value)
    assert T.reshape(4, 4).T.tobytes() == T1.tobytes()
    s.solve(corr, False, 3)                          # without a reset the solve continues from the current pose
    r = o32.picp_solve(OCam(480, 640, 0, 10, fp["K"], np.eye(4)), fp["model"], fp["cur_pts"], corr, 10, 10000.0, False, trace=False)
    assert np.abs(s.camera().worldInCameraPose() - r["T"]).max() < 1e-4
    ctx.free(d_I); s.close()


def test_handles_do_not_leak_device_memory():
    """create / use / destroy contexts, solvers, pipelines, graphs and events repeatedly: free device memory returns
    to where it was.  Measured in a process of its own (tools/leak_probe.py): inside the test process the runtime's
    own pools and other tests' live objects move the number."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "leak_probe.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"^(\w+)\s*: drift MiB after each cycle:(.*)$", r.stdout, flags=re.M)
    assert [k for k, _ in rows] == ["ctx", "event", "match", "frame", "capture", "all"], r.stdout
    for kind, vals in rows:
        drift = [float(x) for x in vals.split()]
        # a handle that is not released grows the figure on EVERY cycle, so the typical (median) step between cycles must
        # be zero.  The HIP runtime's own pools may step up once or twice on the way (measured: usually 0.00 throughout;
        # sometimes one or two steps of 6..52 MiB at some cycle, flat before and after): at most four of the 15 steps may go up
        # (a leaking handle steps up on all of them).
        steps = sorted(b - a for a, b in zip(drift[:-1], drift[1:]))
        n_up = sum(1 for x in steps if x > 0.05)
        assert len(drift) == 16 and abs(steps[len(steps) // 2]) < 0.05 and steps[2] > -0.05 and n_up <= 4 and max(drift) < 192.0, (kind, drift)


@pytest.mark.parametrize("form", [1, 2, 3])
def test_batched_solver_reports_bad_indices(vo, ctx, o32, form):
    """a pair whose index lies outside the point arrays is dropped AND counted (stats_out[4p + 3]) by every form of
    the batched solver -- the single-problem entry points report the same input as VO_ERR_BAD_INDEX"""
    P, n = 3, 1500
    fps = [vo.synth.frame_pair(n, seed=5100 + p) for p in range(P)]
    pairs = [_corr(f).copy() for f in fps]
    pairs[1][[5, 700, 1400], 1] = [n + 7, -3, 2 ** 30]       # three bad model indices in problem 1
    pairs[2][11, 0] = n                                       # one bad measurement index in problem 2
    world = np.stack([f["model"] for f in fps]); meas = np.stack([f["cur_pts"] for f in fps])
    pbuf = np.stack(pairs).astype(np.int32); npairs = np.full(P, n, np.int32)
    d = [ctx.to_device(a) for a in (world, meas, pbuf, npairs)]
    d_T = ctx.alloc(P * 64); d_stats = ctx.alloc(P * 16)
    K = np.ascontiguousarray(fps[0]["K"].T).ravel()
    lib = ctx.lib
    assert lib.vo_picp_batch_set_form(ctx.h, form) == 0
    try:
        rc = lib.vo_picp_solve_batch_dev(ctx.h, P, 480, 640, 0, 10, K.ctypes.data_as(C.c_void_p), C.c_float(10000.0), 0,
                                         C.c_void_p(d[0]), C.c_size_t(n), C.c_void_p(d[1]), C.c_size_t(n), C.c_void_p(d[2]),
                                         C.c_size_t(n), C.c_void_p(d[3]), None, 8, C.c_void_p(d_T), C.c_void_p(d_stats))
        assert rc == 0, lib.vo_last_error()
        T = np.zeros((P, 16), np.float32); st = np.zeros((P, 4), np.float32)
        ctx.d2h(T, d_T); ctx.d2h(st, d_stats)
    finally:
        lib.vo_picp_batch_set_form(ctx.h, 0)
        for x in d + [d_T, d_stats]:
            ctx.free(x)
    assert st[:, 3].tolist() == [0.0, 3.0, 1.0]
    for p in range(P):
        ok = (pairs[p][:, 0] >= 0) & (pairs[p][:, 0] < n) & (pairs[p][:, 1] >= 0) & (pairs[p][:, 1] < n)
        r = o32.picp_solve(OCam(480, 640, 0, 10, fps[p]["K"], np.eye(4)), fps[p]["model"], fps[p]["cur_pts"], pairs[p][ok],
                           8, 10000.0, False, trace=False)
        assert int(st[p, 2]) == r["num_inliers"] and np.abs(T[p].reshape(4, 4).T - r["T"]).max() < 1e-4


def test_contexts_on_concurrent_host_threads(vo, o32):
    """One context per host thread, four threads driving the whole frame (match -> join -> rounds -> triangulate) at the same
    time through the C ABI (ctypes releases the GIL inside every call): the library keeps no state outside its handles, so every
    thread must get exactly the single-threaded result of its own frame."""
    import threading
    fps = [vo.synth.frame_pair(3000 + 500 * k, seed=9100 + k, drop=0.05, distractors=30, model_drop=0.05) for k in range(4)]

    def frame(ctx, fp):
        m = vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=ctx)
        j = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
        s = vo.PICPSolver(ctx)
        s.setKernelThreshold(10000.0)
        s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
        s.solve(j, False, 15)
        T = s.camera().worldInCameraPose().copy()
        n_in = s.numInliers()
        s.close()
        xyz, pairs, app = vo.triangulate_points(fp["K"], T, m, fp["ref_pts"], fp["cur_pts"], fp["cur_app"], ctx=ctx)
        return (m.tobytes(), j.tobytes(), T.tobytes(), n_in, xyz.tobytes(), pairs.tobytes(), app.tobytes())

    c0 = vo.Context(0)
    expect = [frame(c0, fp) for fp in fps]
    c0.close()
    for k, fp in enumerate(fps):                       # anchored on the oracle, not only on itself
        assert np.array_equal(np.frombuffer(expect[k][0], np.int32).reshape(-1, 2), o32.match(fp["ref_app"], fp["cur_app"]))
    errors = []

    def worker(k):
        try:
            ctx = vo.Context(0)
            for _ in range(12):
                got = frame(ctx, fps[k])
                if got != expect[k]:
                    errors.append((k, [a == b for a, b in zip(got, expect[k])]))
                    break
            ctx.close()
        except Exception as e:                        # noqa: BLE001 -- reported through the assertion below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=300)
    assert not any(t.is_alive() for t in threads) and not errors, errors


def test_two_solvers_interleaved_on_one_context(vo, ctx, o32):
    """Two PICPSolver handles of one context used alternately, round by round, each on its own problem and with its own
    correspondences / threshold / arithmetic mode: nothing of one may leak into the other (cached correspondences, captured
    graphs, partial buffers are per handle)."""
    fa = vo.synth.frame_pair(2500, seed=9200, drop=0.05, distractors=10, model_drop=0.05)
    fb = vo.synth.frame_pair(900, seed=9201)
    probs = []
    for fp, thr, keep, exact in ((fa, 10000.0, False, False), (fb, 40.0, True, True)):
        m = vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=ctx)
        j = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
        probs.append((fp, j, thr, keep, exact))

    def make(fp, thr, exact):
        s = vo.PICPSolver(ctx)
        s.setKernelThreshold(thr); s.setExact(exact)
        s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
        return s

    alone = []
    for fp, j, thr, keep, exact in probs:
        s = make(fp, thr, exact)
        T = []
        for _ in range(10):
            s.oneRound(j, keep); T.append(s.camera().worldInCameraPose().tobytes())
        alone.append(T); s.close()
    solvers = [make(fp, thr, exact) for fp, j, thr, keep, exact in probs]
    both = [[], []]
    for _ in range(10):
        for k, (fp, j, thr, keep, exact) in enumerate(probs):
            solvers[k].oneRound(j, keep)
        for k in (1, 0):
            both[k].append(solvers[k].camera().worldInCameraPose().tobytes())
    for s in solvers: s.close()
    assert both == alone
    fp, j, thr, keep, _ = probs[1]                                          # and the exact one is the oracle's, bit for bit
    r = o32.picp_solve_raw(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)), fp["model"], fp["cur_pts"], j, 10, thr, keep)
    assert alone[1][-1] == r["T"][-1].astype(np.float32).tobytes()


@pytest.mark.parametrize("n,keep,thr", [(1, False, 10000.0), (37, False, 10000.0), (127, True, 40.0), (256, False, 10000.0), (200, True, 25.0)])
def test_small_problem_form_equals_the_round_kernels(vo, o32, n, keep, thr):
    """Up to 256 correspondences the solver runs all rounds of a call in ONE launch (picp_small_kernel); VO_PICP_SMALL=0 keeps
    the launch-per-round form.  Same arithmetic by construction: pose, H, b and the statistics must agree bit for bit, for one
    round at a time as well as for a whole solve, and stay within the usual tolerance of the oracle."""
    import subprocess, sys, json
    code = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as g
vo = g.load_package()
n, keep, thr = %d, %s, %r
fp = vo.synth.frame_pair(max(n, 8), seed=9700 + n, noise_px=1.5)
corr = np.stack([fp["gt_matches"][:n, 1], fp["model_pairs"][fp["gt_matches"][:n, 0], 1]], 1).astype(np.int32)
ctx = vo.Context(0)
out = []
for mode in ("solve", "rounds"):
    s = vo.PICPSolver(ctx)
    s.setKernelThreshold(thr)
    s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
    if mode == "solve":
        s.solve(corr, keep, 9)
    else:
        for _ in range(9):
            s.oneRound(corr, keep)
    H, b = s.system()
    out.append([s.camera().worldInCameraPose().tobytes().hex(), H.tobytes().hex(), b.tobytes().hex(),
                float(s.chiInliers()).hex(), float(s.chiOutliers()).hex(), s.numInliers()])
    s.close()
print(json.dumps(out))
''' % (os.path.join(os.path.dirname(__file__), ".."), n, keep, thr)
    res = {}
    for small in ("1", "0"):
        env = dict(os.environ, VO_PICP_SMALL=small)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        res[small] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["1"][0] == res["1"][1]                    # one launch of nine rounds = nine launches of one
    assert res["1"] == res["0"]                          # = the launch-per-round kernels, bit for bit
    # and the oracle, within the fast mode's tolerance
    fp = vo.synth.frame_pair(max(n, 8), seed=9700 + n, noise_px=1.5)
    corr = np.stack([fp["gt_matches"][:n, 1], fp["model_pairs"][fp["gt_matches"][:n, 0], 1]], 1).astype(np.int32)
    r = o32.picp_solve(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)), fp["model"], fp["cur_pts"], corr, 9, thr, keep, trace=False)
    T = np.frombuffer(bytes.fromhex(res["1"][0][0]), np.float32).reshape(4, 4)
    assert res["1"][0][5] == r["num_inliers"]
    if n >= 16:
        assert np.abs(T - r["T"]).max() < 1e-4


def test_batched_frames_are_bitwise_reproducible_run_to_run(vo):
    """The sorted matcher variants rank points inside a bin with LDS atomics (arbitrary order inside a bin), workgroups finish in
    any order: none of that may reach a result.  40 runs of one 16-frame call: every output buffer identical, bit for bit."""
    import hashlib
    c = vo.Context(0)
    fps = [vo.synth.frame_pair(6000, seed=9800 + 3 * k, distractors=40) for k in range(16)]
    bp = vo.BatchPipeline(c, fps, n_iters=12)
    seen = set()
    for _ in range(40):
        bp.run()
        h = hashlib.sha256()
        h.update(bp.counts().tobytes()); h.update(bp.poses().tobytes()); h.update(bp.stats().tobytes())
        for f in (0, 7, 15):
            for what in ("match", "join", "tri_xyz", "tri_pairs", "tri_app"):
                h.update(bp.fetch(what, f).tobytes())
        seen.add(h.hexdigest())
    bp.close(); c.close()
    assert len(seen) == 1


def test_device_arrays_off_an_8_byte_boundary_are_refused(vo, ctx):
    """vo_hip.h, Conventions: rows and index pairs move as 8-byte pieces, so a device array that starts 4 bytes into one is
    an argument error (reported before any launch), not a slow path."""
    lib = ctx.lib
    n = 64
    d_a = ctx.alloc(40 * n + 16)
    d_out = ctx.alloc(8 * n + 16)
    d_cnt = ctx.alloc(16)
    try:
        ok = lib.vo_match_appearances_dev(ctx.h, C.c_void_p(d_a), C.c_int(n), C.c_void_p(d_a), C.c_int(n), C.c_float(0.1),
                                          C.c_void_p(d_out), C.c_void_p(d_cnt))
        assert ok == 0
        ctx.synchronize()
        for a1, out in ((d_a + 4, d_out), (d_a, d_out + 4)):
            rc = lib.vo_match_appearances_dev(ctx.h, C.c_void_p(a1), C.c_int(n), C.c_void_p(d_a), C.c_int(n), C.c_float(0.1),
                                              C.c_void_p(out), C.c_void_p(d_cnt))
            assert rc == -1                                            # VO_ERR_INVALID_ARG
            assert b"8-byte" in lib.vo_last_error()
    finally:
        for p in (d_a, d_out, d_cnt):
            ctx.free(p)


def test_single_pass_compaction_equals_the_default(vo, ctx):
    """VO_ONE_PASS=1 (geom.hip: chained scan with decoupled look-back -- built in round 5, measured slower, kept opt-in):
    matcher output, join (+ the solver's gather), triangulation of a batched call and of single-frame calls, bit for bit
    what the count / scan / scatter form writes, survivors in the reference's order (utils.cpp:97-99,126-128)."""
    import os
    rng = np.random.default_rng(5)
    batch = []
    for p in range(9):                        # equal sizes (a batch needs them); the gaps are planted afterwards
        fp = dict(vo.synth.frame_pair(9000, seed=5100 + p))
        n = len(fp["cur_app"])
        fp["cur_app"] = fp["cur_app"].copy(); fp["model_pairs"] = fp["model_pairs"].copy()
        lost = rng.choice(n, n // 10, replace=False)
        fp["cur_app"][lost] = rng.uniform(-1, 1, (len(lost), 10)).astype(np.float32)       # queries without a match
        if p % 2:   # no bitwise copies at all: every key goes through the search and the matcher's own compaction
            fp["cur_app"] = (fp["cur_app"] + rng.normal(0, 0.003, fp["cur_app"].shape)).astype(np.float32)
        fp["model_pairs"][rng.choice(len(fp["model_pairs"]), len(fp["model_pairs"]) // 5, replace=False), 0] = -1   # no partner
        batch.append(fp)
    singles = [vo.synth.frame_pair(7000, seed=5200 + p, drop=0.1, distractors=50, model_drop=0.2) for p in range(3)]
    res = {}
    for one_pass in ("0", "1"):
        os.environ["VO_ONE_PASS"] = one_pass
        try:
            out = []
            bp = vo.BatchPipeline(ctx, batch, n_iters=4)
            bp.run(); ctx.synchronize()
            c = bp.counts()
            assert np.all(c[0] < 9000) and np.all(c[1] < c[0]) and np.all(c[2] > 0) and np.all(c[2] <= c[0])
            out.append((c.tobytes(), bp.poses().tobytes()))
            for f in range(len(batch)):
                out.append(tuple(bp.fetch(w, f).tobytes() for w in ("match", "join", "tri_xyz", "tri_pairs", "tri_app")))
            bp.close()
            for fp in singles:
                m = vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=ctx)
                j = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
                xyz, pairs, app = vo.triangulate_points(fp["K"], fp["X_gt"], m, fp["ref_pts"], fp["cur_pts"], fp["cur_app"], ctx=ctx)
                assert 0 < len(xyz) <= len(m) and 0 < len(j) < len(m)
                out.append((m.tobytes(), j.tobytes(), xyz.tobytes(), pairs.tobytes(), app.tobytes()))
            res[one_pass] = out
        finally:
            os.environ.pop("VO_ONE_PASS", None)
    assert res["0"] == res["1"]


@pytest.mark.parametrize("general_k", [False, True])
def test_batched_forms_agree_on_nan_and_inf_world_points(vo, ctx, o32, general_k):
    """the one-workgroup-per-problem kernel multiplies its Jacobian terms with v_mul_legacy_f32 (0 * anything = 0,
    vo_math.h: vo_mul0), the launch-per-round form with plain products and selects.  A world point that really is NaN (not
    the DROPPED marker) passes every gate of the reference (all comparisons false, camera.h:28-35) and poisons H, b and the
    pose -- in both forms, as in the oracle; so does x = -inf (R p at the identity holds 0 * inf = NaN in pc.y and pc.z).  A
    point at z = +-inf fails the depth gate (pc.z beyond z_far / z_near: a true comparison) and must leave no trace -- in
    both forms, although its pc.x is NaN: a rejected term contributes exact zeros."""
    import ctypes as C
    from oracle.oracle import Camera as OCam
    fp = vo.synth.frame_pair(3000, seed=61)
    K = fp["K"].copy()
    if general_k:
        K = np.array([[180.0, 0.7, 320.0], [0.0, 175.0, 240.0], [0.0, 0.0, 1.0]], np.float32)
    n = len(fp["model"])
    gt = fp["gt_matches"]
    pairs = np.stack([gt[:, 1], gt[:, 0]], 1).astype(np.int32)              # (cur idx, model idx): model i <-> ref i here
    P = 5
    worlds = np.stack([fp["model"]] * P).astype(np.float32)
    worlds[1, 100] = np.nan                                                 # poison
    worlds[2, 200, 2] = np.inf                                              # rejected by the depth gate
    worlds[3, 300, 2] = -np.inf                                             # rejected by the depth gate
    worlds[4, 400, 0] = -np.inf                                             # poison (0 * inf in pc.y, pc.z)
    poisoned = (1, 4)
    d_w, d_m = ctx.to_device(worlds), ctx.to_device(np.stack([fp["cur_pts"]] * P))
    d_p, d_n = ctx.to_device(np.stack([pairs] * P)), ctx.to_device(np.full(P, n, np.int32))
    d_T, d_s = ctx.alloc(P * 64), ctx.alloc(P * 16)
    Kc = np.ascontiguousarray(K.T).ravel()
    out = {}
    try:
        for form in (1, 2):
            assert ctx.lib.vo_picp_batch_set_form(ctx.h, form) == 0
            rc = ctx.lib.vo_picp_solve_batch_dev(ctx.h, P, 480, 640, 0, 10, Kc.ctypes.data_as(C.c_void_p), C.c_float(10000.0), 0,
                                                 C.c_void_p(d_w), C.c_size_t(n), C.c_void_p(d_m), C.c_size_t(len(fp["cur_pts"])),
                                                 C.c_void_p(d_p), C.c_size_t(n), C.c_void_p(d_n), None, 8, C.c_void_p(d_T), C.c_void_p(d_s))
            assert rc == 0, ctx.lib.vo_last_error()
            T = np.zeros((P, 16), np.float32); st = np.zeros((P, 4), np.float32)
            ctx.d2h(T, d_T); ctx.d2h(st, d_s)
            out[form] = (T, st)
    finally:
        ctx.lib.vo_picp_batch_set_form(ctx.h, 0)
        for d in (d_w, d_m, d_p, d_n, d_T, d_s):
            ctx.free(d)
    for p in range(P):
        r = o32.picp_solve(OCam(480, 640, 0, 10, K, np.eye(4)), worlds[p], fp["cur_pts"], pairs, 8, 10000.0, False, trace=False)
        assert np.isfinite(r["T"]).all() == (p not in poisoned)             # the reference's own behaviour, as restated
        for form in (1, 2):
            T, st = out[form]
            if p in poisoned:
                assert np.isnan(T[p]).any()
            else:
                assert np.abs(T[p].reshape(4, 4).T - r["T"]).max() < 1e-4 and int(st[p, 2]) == r["num_inliers"] == (n if p == 0 else n - 1)
    for p in (0, 2, 3):                                                     # the two forms differ by the order of their sums only
        assert np.abs(out[1][0][p] - out[2][0][p]).max() < 1e-5 and np.array_equal(out[1][1][p, 1:], out[2][1][p, 1:])
