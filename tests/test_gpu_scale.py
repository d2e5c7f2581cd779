"""Full-size (BASELINE config 2: 50 000 points) checks through size-independent
properties plus a direct oracle comparison where the oracle finishes in seconds."""
import numpy as np
import pytest

from conftest import rel_err
from oracle.oracle import Camera as OCam

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(vo):
    return vo.synth.frame_pair(50000, seed=2000)


def test_matcher_50k_is_the_permutation(vo, ctx, big):
    m = vo.compute_correspondences_images(big["ref_app"], big["cur_app"], ctx=ctx)
    gt = big["gt_matches"]
    gt = gt[np.argsort(gt[:, 1], kind="stable")]
    assert np.array_equal(m, gt)                    # every landmark found, exact indices, query order
    # symmetric call: roles swap, pairs stay (a1 idx, a2 idx)
    m2 = vo.compute_correspondences_images(big["cur_app"], big["ref_app"], ctx=ctx)
    assert np.array_equal(m2[np.argsort(m2[:, 1], kind="stable")][:, ::-1], gt[np.argsort(gt[:, 0], kind="stable")])


def test_frame_50k_against_oracle(vo, ctx, o32, o64, big):
    m = vo.compute_correspondences_images(big["ref_app"], big["cur_app"], ctx=ctx)
    j = vo.extract_correspondences_world(m, big["model_pairs"], ctx=ctx)
    assert np.array_equal(j, o32.join(m, big["model_pairs"], linear=True)) and len(j) == 50000
    cam = vo.Camera(480, 640, 0, 10, big["K"], np.eye(4), ctx=ctx)
    s = vo.PICPSolver(ctx)
    s.setKernelThreshold(10000.0)
    s.init(cam, big["model"], big["cur_pts"])
    s.oneRound(j, False)
    H, b = s.system()
    ocam = OCam(480, 640, 0, 10, big["K"], np.eye(4))
    r64 = o64.picp_solve(ocam, big["model"], big["cur_pts"], j, 50, 10000.0, False)
    r32 = o32.picp_solve(ocam, big["model"], big["cur_pts"], j, 50, 10000.0, False)
    assert rel_err(H - np.eye(6, dtype=np.float32), r64["H"][0]) < 1e-5
    assert rel_err(b, r64["b"][0]) < 1e-5
    # the GPU's tree reduction must be at least as close to ref64 as the reference's own sequential sum
    assert rel_err(H - np.eye(6, dtype=np.float32), r64["H"][0]) <= rel_err(r32["H"][0], r64["H"][0]) * 2 + 1e-6
    s.solve(j, False, 49)
    T = s.camera().worldInCameraPose()
    assert s.numInliers() == 50000 == r32["num_inliers"]
    assert np.abs(T - r32["T"]).max() < 1e-4 and np.abs(T - r64["T"]).max() < 1e-4
    assert np.abs(T - big["X_gt"]).max() < 1e-3                        # and it is the right answer
    assert abs(s.chiInliers() - r64["chi_inliers"]) < 1e-4 * r64["chi_inliers"]
    # triangulate with the estimated pose: survivors exact, points within tolerance, pairs (cur idx, slot)
    xyz, pairs, app = vo.triangulate_points(big["K"], T, m, big["ref_pts"], big["cur_pts"], big["cur_app"], ctx=ctx)
    xo, po, ao = o32.triangulate(big["K"], T, m, big["ref_pts"], big["cur_pts"], big["cur_app"])
    assert np.array_equal(pairs, po) and np.array_equal(app, ao)
    assert np.array_equal(xyz, xo)                            # same pose in: bit for bit
    assert np.array_equal(pairs[:, 1], np.arange(len(pairs)))
    # transform round trip (linearity / invertibility property)
    Xi = np.linalg.inv(T.astype(np.float64)).astype(np.float32)
    back = vo.transform_points(Xi, vo.transform_points(T, big["model"], ctx=ctx), ctx=ctx)
    assert np.abs(back - big["model"]).max() < 1e-4
    s.close()


@pytest.mark.parametrize("form", [1, 2])
def test_batched_solver_matches_single(vo, ctx, form):
    """vo_picp_solve_batch_dev on P problems == P independent PICPSolver runs, in both forms (1: one launch per
    round, the single-problem kernels with the problem as a grid dimension; 2: one workgroup per problem)."""
    import ctypes as C
    assert ctx.lib.vo_picp_batch_set_form(ctx.h, form) == 0
    P, n, iters = 5, 3000, 15
    fps = [vo.synth.frame_pair(n, seed=4000 + p) for p in range(P)]
    lib = ctx.lib
    world = np.stack([f["model"] for f in fps]); meas = np.stack([f["cur_pts"] for f in fps])
    pairs = []
    for f in fps:
        mp = dict(f["model_pairs"].tolist())
        pairs.append(np.array([(c, mp[r]) for r, c in f["gt_matches"].tolist()], np.int32))
    npairs = np.array([len(p) for p in pairs], np.int32)
    stride = max(npairs)
    pbuf = np.zeros((P, stride, 2), np.int32)
    for i, p in enumerate(pairs):
        pbuf[i, : len(p)] = p
    d_world, d_meas, d_pairs, d_n = (ctx.to_device(a) for a in (world, meas, pbuf, npairs))
    d_T = ctx.alloc(P * 16 * 4); d_stats = ctx.alloc(P * 4 * 4)
    K = np.ascontiguousarray(fps[0]["K"].T).ravel()
    rc = lib.vo_picp_solve_batch_dev(ctx.h, P, 480, 640, 0, 10, K.ctypes.data_as(C.c_void_p), C.c_float(10000.0), 0,
                                     C.c_void_p(d_world), C.c_size_t(n), C.c_void_p(d_meas), C.c_size_t(n),
                                     C.c_void_p(d_pairs), C.c_size_t(stride), C.c_void_p(d_n), None, iters,
                                     C.c_void_p(d_T), C.c_void_p(d_stats))
    assert rc == 0, lib.vo_last_error()
    T = np.zeros((P, 16), np.float32); st = np.zeros((P, 4), np.float32)
    ctx.d2h(T, d_T); ctx.d2h(st, d_stats)
    for p in range(P):
        cam = vo.Camera(480, 640, 0, 10, fps[p]["K"], np.eye(4), ctx=ctx)
        s = vo.PICPSolver(ctx); s.setKernelThreshold(10000.0)
        s.init(cam, fps[p]["model"], fps[p]["cur_pts"])
        s.solve(pairs[p], False, iters)
        Ts = s.camera().worldInCameraPose()
        assert np.abs(T[p].reshape(4, 4).T - Ts).max() < 2e-5          # different reduction tree only
        assert int(st[p, 2]) == s.numInliers() == n
        assert np.abs(T[p].reshape(4, 4).T - fps[p]["X_gt"]).max() < 1e-3
        s.close()
    for d in (d_world, d_meas, d_pairs, d_n, d_T, d_stats):
        ctx.free(d)
    ctx.lib.vo_picp_batch_set_form(ctx.h, 0)


def test_beyond_benchmark_size_300k(vo, ctx, o32, o64):
    """6x the benchmark size: the solver's grid is capped (4 workgroups per CU) so every thread takes
    several correspondences, the matcher's bucket sort and the compactions run with multi-block scans."""
    n = 300000
    fp = vo.synth.frame_pair(n, seed=2100)
    m = vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=ctx)
    gt = fp["gt_matches"]
    assert np.array_equal(m, gt[np.argsort(gt[:, 1], kind="stable")])
    j = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
    assert np.array_equal(j, o32.join(m, fp["model_pairs"], linear=True)) and len(j) == n
    cam = vo.Camera(480, 640, 0, 10, fp["K"], np.eye(4), ctx=ctx)
    s = vo.PICPSolver(ctx)
    s.setKernelThreshold(10000.0)
    s.init(cam, fp["model"], fp["cur_pts"])
    s.solve(j, False, 6)
    T = s.camera().worldInCameraPose()
    ocam = OCam(480, 640, 0, 10, fp["K"], np.eye(4))
    r64 = o64.picp_solve(ocam, fp["model"], fp["cur_pts"], j, 6, 10000.0, False, trace=False)
    assert s.numInliers() == n == r64["num_inliers"]
    assert np.abs(T - r64["T"]).max() < 1e-4
    assert abs(s.chiInliers() - r64["chi_inliers"]) < 1e-4 * r64["chi_inliers"]
    xyz, pairs, _ = vo.triangulate_points(fp["K"], T, m, fp["ref_pts"], fp["cur_pts"], ctx=ctx)
    xo, po, _ = o32.triangulate(fp["K"], T, m, fp["ref_pts"], fp["cur_pts"])
    assert np.array_equal(pairs, po) and np.array_equal(xyz, xo)
    s.close()


@pytest.mark.parametrize("n,drop", [(70000, 0.05), (100000, 0.0)])
def test_cell_hash_matcher_beyond_the_staging_budgets(vo, o32, n, drop):
    """matcher mode 3 on sets larger than the LDS budgets of its placement (more level-1 workgroups at 70k; at 100k a
    slice no longer fits and the placement writes straight) -- against the reference's kd-tree restatement"""
    c = vo.Context(0)
    assert c.lib.vo_match_set_mode(c.h, 3) == 0
    fp = vo.synth.frame_pair(n, seed=77, drop=drop, distractors=200 if drop else 0)
    m = vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=c)
    m_o = o32.match_kdtree(fp["ref_app"], fp["cur_app"])
    assert np.array_equal(m, m_o) and len(m) > 0.85 * n
    c.close()


@pytest.mark.parametrize("dist", ["cluster", "line", "heavy_tail", "duplicates"])
def test_pruned_searches_equal_the_full_scan_on_hostile_distributions(vo, dist):
    """Distributions that defeat the bucketing (a coarse bin holding more points than a 16-bit relative start or the LDS
    staging can take, all spread in one component, outliers stretching the grid, thousands of exact duplicates): the pruned
    searches fall back to their global-memory walks and must still return the pairs of the full scan (mode 1, itself checked
    against the oracle elsewhere), exact ties to the lowest index included."""
    rng = np.random.default_rng(11)
    n = 80000
    if dist == "cluster":
        a = np.concatenate([rng.normal(0.0, 0.01, (70000, 10)), rng.uniform(-1, 1, (n - 70000, 10))])
    elif dist == "line":
        a = np.zeros((n, 10)); a[:, 7] = rng.uniform(-50, 50, n); a += rng.normal(0, 1e-3, (n, 10))
    elif dist == "heavy_tail":
        a = rng.standard_cauchy((n, 10)) * 0.05
    else:
        base = rng.uniform(-1, 1, (4000, 10)); a = base[rng.integers(0, 4000, n)]
    a = a.astype(np.float32)
    perm = rng.permutation(n)
    b = (a[perm].astype(np.float64) + (0 if dist == "duplicates" else rng.normal(0, 2e-3, (n, 10)))).astype(np.float32)
    b = b[: n - 5000]                                                     # unequal set sizes: the larger one is searched
    res = []
    for mode in (1, 2, 3):
        c = vo.Context(0)
        assert c.lib.vo_match_set_mode(c.h, mode) == 0
        res.append(vo.compute_correspondences_images(a, b, ctx=c))
        c.close()
    assert len(res[0]) > 0.5 * len(b)
    assert np.array_equal(res[1], res[0]) and np.array_equal(res[2], res[0])
