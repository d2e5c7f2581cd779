"""BASELINE configs[2] and [3] at their FULL sizes inside the -m gpu suite.

 * config 4's per-GPU share: ONE vo_frames_batch_dev call over 200 frame pairs x 50 000 points x 50 rounds
   (gridDim over 200 frames, 200 MB of packed correspondences, the cell-hash matcher picked by the auto rule): every
   frame against the generator's ground truth, three frames stage by stage against the oracle (its matcher is the
   reference's own PCA kd-tree: the double loop would take minutes).
 * config 3: the device-resident sequence at its full 200 frames x ~50k landmarks in view x 100 rounds, solver in reference-order
   arithmetic: every count and every pose of the chain against the oracle-side run of the same loop, bit for bit.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import vo_pipeline as P
from oracle.oracle import Camera as OCam

pytestmark = pytest.mark.gpu
N, F, ITERS = 50000, 200, 50


def test_batched_frames_200x50k(vo, ctx, o32):
    distinct = [vo.synth.frame_pair(N, seed=4000 + p) for p in range(8)]        # config 4's seeds; values repeat every
    fps = [distinct[i % 8] for i in range(F)]                                  # 8 frames, all 200 are separate copies
    bp = vo.BatchPipeline(ctx, lambda lo, hi: fps[lo:hi], n_iters=ITERS, n_frames=F, upload_block=40)
    bp.run()
    ctx.synchronize()
    c, T, st = bp.counts(), bp.poses(), bp.stats()
    # every frame: all matches, all joins, all inliers, the generator's pose
    assert np.all(c[0] == N) and np.all(c[1] == N) and np.all(st[:, 2] == N)
    assert np.all(c[2] > 0.5 * N) and np.all(c[2] <= N)
    err = np.abs(T - bp.X_gt).reshape(F, -1).max(axis=1)
    assert err.max() < 1e-3, (int(err.argmax()), float(err.max()))
    for f in range(F):                                                         # copies of one pair: identical results
        assert np.array_equal(T[f], T[f % 8]) and np.array_equal(c[:, f], c[:, f % 8])
    # three frames stage by stage against the oracle
    for f in (0, 101, 199):
        fp = fps[f]
        m = bp.fetch("match", f)
        m_o = o32.match_kdtree(fp["ref_app"], fp["cur_app"])
        assert np.array_equal(m, m_o)
        assert np.array_equal(m, fp["gt_matches"])                             # = the generator's permutation
        j = bp.fetch("join", f)
        j_o = o32.join(m_o, fp["model_pairs"], linear=True)
        assert np.array_equal(j, j_o)
        r = o32.picp_solve(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)), fp["model"],
                           fp["cur_pts"], j_o, ITERS, 10000.0, False, trace=False)
        assert np.abs(T[f] - r["T"]).max() < 1e-4 and int(st[f, 2]) == r["num_inliers"]
        assert abs(st[f, 0] - r["chi_inliers"]) <= 1e-4 * r["chi_inliers"]
        xyz, pairs, app = bp.fetch("tri_xyz", f), bp.fetch("tri_pairs", f), bp.fetch("tri_app", f)
        xo, po, ao = o32.triangulate(fp["K"], T[f], m_o, fp["ref_pts"], fp["cur_pts"], fp["cur_app"])
        assert np.array_equal(pairs, po) and np.array_equal(app, ao)
        assert np.array_equal(xyz, xo)                                         # same pose in, same operations: bit for bit
    bp.close()


def test_batched_frames_200x50k_without_bitwise_copies(vo, ctx, o32):
    """The same call when NO current descriptor is a bitwise copy of its landmark's (a front end that recomputes descriptors
    per frame): current rows = reference rows + N(0, 0.01^2) per component, so every query is answered by the nearest-neighbour
    SEARCH (vo_complete.cpp:12-49, brute_force_search.h:22-41) and none by the exact-duplicate pass.  All 200 frames against
    the generator's permutation, three of them against the oracle's kd-tree matcher and through the rest of the frame."""
    rng = np.random.default_rng(77)
    distinct = []
    for p in range(8):
        fp = dict(vo.synth.frame_pair(N, seed=4000 + p))
        fp["cur_app"] = (fp["cur_app"] + rng.normal(0.0, 0.01, fp["cur_app"].shape)).astype(np.float32)
        assert not (fp["cur_app"][fp["gt_matches"][:, 1]] == fp["ref_app"][fp["gt_matches"][:, 0]]).all(axis=1).any()
        distinct.append(fp)
    fps = [distinct[i % 8] for i in range(F)]
    bp = vo.BatchPipeline(ctx, lambda lo, hi: fps[lo:hi], n_iters=10, n_frames=F, upload_block=40)
    for mode in (0, 3, 5):                                  # automatic, the search alone, exact-duplicate pass + search
        assert ctx.lib.vo_match_set_mode(ctx.h, mode) == 0
        try:
            bp.match_only()
            ctx.synchronize()
        finally:
            assert ctx.lib.vo_match_set_mode(ctx.h, 0) == 0
        assert np.all(bp.counts()[0] == N)
        for f in range(F):
            assert np.array_equal(bp.fetch("match", f), fps[f]["gt_matches"]), (mode, f)
    bp.run()
    ctx.synchronize()
    c, T, st = bp.counts(), bp.poses(), bp.stats()
    assert np.all(c[0] == N) and np.all(c[1] == N) and np.all(st[:, 2] == N)
    for f in (0, 101, 199):
        fp = fps[f]
        m = bp.fetch("match", f)
        m_o = o32.match_kdtree(fp["ref_app"], fp["cur_app"])
        assert np.array_equal(m, m_o)
        assert np.array_equal(bp.fetch("join", f), o32.join(m_o, fp["model_pairs"], linear=True))
        xyz, pairs, app = bp.fetch("tri_xyz", f), bp.fetch("tri_pairs", f), bp.fetch("tri_app", f)
        xo, po, ao = o32.triangulate(fp["K"], T[f], m_o, fp["ref_pts"], fp["cur_pts"], fp["cur_app"])
        assert np.array_equal(pairs, po) and np.array_equal(app, ao) and np.array_equal(xyz, xo)
    bp.close()


@pytest.mark.parametrize("n,f", [(6000, 12), (N, 24)])
def test_batched_frames_reference_order_form(vo, ctx, o32, n, f):
    """the same call with the solver stage in reference-order arithmetic (vo_picp_batch_set_form(3)): poses and
    statistics of every frame equal ref32 bit for bit -- 12 frames x 6 000 points, and 24 frames x 50 000 points x 50 rounds
    (one workgroup per frame, 0.18 ms per round)"""
    fps = [vo.synth.frame_pair(n, seed=4100 + p) for p in range(f)]
    assert ctx.lib.vo_picp_batch_set_form(ctx.h, 3) == 0
    try:
        bp = vo.BatchPipeline(ctx, fps, n_iters=ITERS if n == N else 20)
        bp.run()
        T, st = bp.poses(), bp.stats()
        for i, fp in enumerate(fps):
            j = bp.fetch("join", i)
            if n == N:
                assert len(j) == n                       # (the matcher and the join are checked against the oracle elsewhere)
            else:
                assert np.array_equal(j, o32.join(o32.match(fp["ref_app"], fp["cur_app"]), fp["model_pairs"]))
            r = o32.picp_solve(OCam(480, 640, 0, 10, fp["K"], np.eye(4)), fp["model"], fp["cur_pts"], j, ITERS if n == N else 20, 10000.0, False,
                               trace=False)
            assert np.array_equal(T[i], r["T"].astype(np.float32)), i
            assert st[i, 0] == np.float32(r["chi_inliers"]) and st[i, 1] == np.float32(r["chi_outliers"]) and int(st[i, 2]) == r["num_inliers"]
        bp.close()
    finally:
        assert ctx.lib.vo_picp_batch_set_form(ctx.h, 0) == 0


@pytest.mark.parametrize("n_frames", [200])            # BASELINE config 3 at its full length (40 frames until the reference-order solver got 5x faster)
def test_sequence_200x50k_is_bit_identical_to_ref32(vo, ctx, o32, n_frames):
    seq = vo.synth.sequence(seed=3000, n_frames=n_frames, n_visible=N)
    n = [len(f["pts"]) for f in seq["frames"]]
    assert min(n) > 45000
    sp = vo.SequencePipeline(ctx, seq, n_iters=100, keep_map=True)      # the loop body's map upkeep inside the chain, on the device
    assert sp.lib.vo_picp_set_exact(sp.solver, 1) == 0
    sp.run()
    traj, counts = sp.trajectory(), sp.counts()
    n_in = sp.stats()[2]
    map_pts, map_app = sp.map.read()
    sp.close()
    frames = [(f["pts"], f["app"]) for f in seq["frames"]]
    res = P.run_sequence(frames, seq["K"], seq["rows"], seq["cols"], seq["z_near"], seq["z_far"], rounds=100, o=o32,
                         X0=traj[1], kdtree=True, keep_map=True)
    # the first relative pose at the path's scale: vo_estimate_transform_dev (sums over ~50k correspondences on the GPU, epi.hip)
    # against the oracle's numpy restatement of epipolar_utils.cpp:103-213 on the same pair
    f0, f1 = seq["frames"][0], seq["frames"][1]
    corr01 = o32.match_kdtree(f0["app"], f1["app"])
    X0_o = P.estimate_transform(o32, seq["K"], corr01, f0["pts"], f1["pts"])
    assert len(corr01) > 40000 and np.abs(np.asarray(traj[1]) - X0_o).max() < 5e-5, np.abs(np.asarray(traj[1]) - X0_o).max()
    # the map at the path's scale (vo_complete.cpp:145-147,175-176; PointCloud.h:52-66): ~0.8 M entries after 200 frames of ~50k
    # points, every entry in the reference's order, points and appearance bits
    want = res["map"]
    assert len(map_pts) == len(want.pts) > 10 * N
    assert map_pts.tobytes() == np.array(want.pts, np.float32).reshape(-1, 3).tobytes()
    assert map_app.tobytes() == np.array(want.app, np.float32).reshape(-1, 10).tobytes()
    exp = np.array(res["stats"], dtype=int)
    assert np.array_equal(counts[2:, 0], exp[:, 0]) and np.array_equal(counts[2:, 1], exp[:, 1])    # matches, joins
    assert np.array_equal(counts[1:, 2], np.array(res["tri_counts"]))                              # triangulated points
    assert n_in == exp[-1, 2] and exp[:, 1].min() > 35000
    assert np.array_equal(np.array(traj), np.array(res["trajectory"], dtype=np.float32))           # every pose, bit for bit
    # and the default (fast) arithmetic stays within rounding of it
    sp = vo.SequencePipeline(ctx, seq, n_iters=100)
    sp.run()
    t_fast, c_fast = sp.trajectory(), sp.counts()
    sp.close()
    assert np.array_equal(c_fast[:, 0], counts[:, 0])                       # matches do not depend on the pose
    # joined pairs follow the previous frame's triangulation, whose cheirality test sees a pose that differs in the
    # last bits: a borderline point may flip
    assert np.abs(c_fast[:, 1:] - counts[:, 1:]).max() <= 8
    # all 199 pairs matched up front by one batched call (cell-hash search at this size): the same chain, bit for bit
    sp = vo.SequencePipeline(ctx, seq, n_iters=100, prematch=True)
    sp.run()
    assert np.array_equal(sp.counts(), c_fast) and np.array_equal(sp.trajectory(), t_fast)
    sp.close()
    # rounding-level differences of one frame's pose scale the next frame's model: over 200 chained frames they grow to ~2e-3
    assert np.abs(np.array(t_fast) - np.array(traj)).max() < 5e-3 and np.abs(np.array(t_fast[:40]) - np.array(traj[:40])).max() < 5e-4


def test_properties_at_full_size(vo, ctx, o32):
    """Size-independent properties of the path at 50 000 points (the oracle's double loops would take minutes here):
    permutation equivariance and role symmetry of the matcher, join with the identity, transform round trip, triangulate ->
    project consistency, solver independence of the correspondence order, reference-order solver = its own batched form."""
    fp = vo.synth.frame_pair(N, seed=2000)
    rng = np.random.default_rng(21)
    a, b = fp["ref_app"], fp["cur_app"]
    m = vo.compute_correspondences_images(a, b, ctx=ctx)
    assert np.array_equal(m, fp["gt_matches"])                                      # the generator's permutation
    # (1) permuting the queries permutes the answer; swapping the roles swaps the columns (equal sizes: either set may be "the tree")
    pq = rng.permutation(N)
    mb = vo.compute_correspondences_images(a, b[pq], ctx=ctx)
    inv = np.empty(N, np.int64); inv[pq] = np.arange(N)
    back = np.stack([m[:, 0], inv[m[:, 1]]], 1)
    assert np.array_equal(mb[np.argsort(mb[:, 1], kind="stable")], back[np.argsort(back[:, 1], kind="stable")].astype(np.int32))
    ms = vo.compute_correspondences_images(b, a, ctx=ctx)
    assert np.array_equal(ms[np.argsort(ms[:, 1], kind="stable")][:, ::-1], m[np.argsort(m[:, 0], kind="stable")])
    # (2) joining with the identity model (ref i <-> model i) renames nothing
    ident = np.stack([np.arange(N), np.arange(N)], 1).astype(np.int32)
    j = vo.extract_correspondences_world(m, ident, ctx=ctx)
    assert np.array_equal(j, np.stack([m[:, 1], m[:, 0]], 1))
    # (3) X^-1 (X p) = p up to two roundings
    X = fp["X_gt"].astype(np.float64)
    Xi = np.linalg.inv(X).astype(np.float32)
    rt = vo.transform_points(Xi, vo.transform_points(fp["X_gt"], fp["model"], ctx=ctx), ctx=ctx)
    assert np.abs(rt - fp["model"]).max() < 2e-5
    # (4) what triangulates from the two views projects back onto the measurements of the second view
    xyz, pairs, _ = vo.triangulate_points(fp["K"], fp["X_gt"], m, fp["ref_pts"], fp["cur_pts"], ctx=ctx)
    assert len(xyz) > 0.97 * N                                       # a few rays diverge under 0.5 px of noise: cheirality reject
    cam = vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], fp["X_gt"], ctx=ctx)   # the points live in the FIRST camera's frame
    uv, n_in = cam.projectPoints(xyz, keep_indices=True)
    ok = uv[:, 0] >= 0
    assert ok.mean() > 0.9 and np.median(np.abs(uv[ok] - fp["cur_pts"][pairs[ok, 0]])) < 1.0       # 0.5 px of noise on both views, baseline 0.1: depth is loose, the reprojection is not
    # (5) the fast solver does not care about the order of the correspondences beyond rounding; the exact one is its batched form
    jw = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
    poses = []
    for order in (np.arange(len(jw)), rng.permutation(len(jw))):
        s = vo.PICPSolver(ctx)
        s.setKernelThreshold(10000.0)
        s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
        s.solve(np.ascontiguousarray(jw[order]), False, 30)
        poses.append(s.camera().worldInCameraPose().copy()); n_in_s = s.numInliers(); s.close()
    assert np.abs(poses[0] - poses[1]).max() < 5e-6 and n_in_s == N and np.abs(poses[0] - fp["X_gt"]).max() < 1e-4
    s = vo.PICPSolver(ctx)
    s.setExact(True); s.setKernelThreshold(10000.0)
    s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
    s.solve(jw, False, 6)
    T_exact = s.camera().worldInCameraPose().copy(); s.close()
    r = o32.picp_solve_raw(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)), fp["model"], fp["cur_pts"], jw, 6, 10000.0, False)
    assert np.array_equal(T_exact, r["T"][-1])                      # 6 rounds x 50 000 sequential adds per accumulator: same bits
