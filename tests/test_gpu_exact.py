"""Reference-order ("exact") solver mode: vo_picp_set_exact / vo_picp_batch_set_form(3).

The per-correspondence terms are unfused, H / b / chi are summed sequentially in correspondence order, the tail is
Eigen's pivoted LDLT with true divisions and double sin/cos -- so everything the solver holds after a round must equal
the float32 oracle (ref32) BIT FOR BIT: H (damping included), b, chi_inliers, chi_outliers, the inlier count, the pose.
On the reference's own dataset (tests/golden/example_data) the whole chain of 119 frames is then bit-identical to the
oracle-side run of the same loop: every count and every pose.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import vo_pipeline as vp
from oracle.oracle import Camera as OCam

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "apps", "bin")
DATA = os.path.join(GOLD, "example_data", "data")


def gold(name):
    d = dict(np.load(os.path.join(GOLD, name)))
    r, c, zn, zf = d["cam_ints"].tolist()
    d.update(rows=r, cols=c, z_near=zn, z_far=zf)
    return d


def exact_trace(vo, ctx, g, world, meas, corr, n_iters, thr, keep, T0=None):
    cam = vo.Camera(g["rows"], g["cols"], g["z_near"], g["z_far"], g["K"], np.eye(4) if T0 is None else T0, ctx=ctx)
    s = vo.PICPSolver(ctx)
    s.setExact(True)
    s.setKernelThreshold(thr)
    s.init(cam, world, meas)
    H, b, st, T = [], [], [], []
    for _ in range(n_iters):
        s.oneRound(corr, keep)
        Hk, bk = s.system()
        H.append(Hk); b.append(bk)
        st.append((s.chiInliers(), s.chiOutliers(), s.numInliers()))
        T.append(s.camera().worldInCameraPose())
    s.close()
    return np.array(H), np.array(b), np.array(st, dtype=np.float32), np.array(T)


@pytest.mark.parametrize("name", ["frame64.npz", "frame1000.npz"])
@pytest.mark.parametrize("thr,keep", [(10000.0, False), (60.0, False), (60.0, True)])
def test_exact_rounds_equal_ref32_bitwise(vo, ctx, o32, name, thr, keep):
    g = gold(name)
    j = g["exp_join"]
    n_it = 12
    r = o32.picp_solve_raw(OCam(g["rows"], g["cols"], g["z_near"], g["z_far"], g["K"], np.eye(4)), g["model"], g["cur_pts"],
                           j, n_it, thr, keep)
    H, b, st, T = exact_trace(vo, ctx, g, g["model"], g["cur_pts"], j, n_it, thr, keep)
    assert np.array_equal(H, r["H"])            # damping included, as _H is left (picp_solver.cpp:102)
    assert np.array_equal(b, r["b"])
    assert np.array_equal(st, r["stats"])       # chi_inliers, chi_outliers (float sums), inlier count
    assert np.array_equal(T, r["T"])
    if thr < 100:
        assert 0 < st[0, 2] < len(j) and st[0, 1] > 0   # the chi test took both branches


def test_exact_picp_test_scenario_100_rounds(vo, ctx, o32):
    """the reference's picp_test scenario (picp_solver_test.cpp:42-79) from a non-identity start: 100 chained rounds,
    one launch (solve) == 100 x oneRound == ref32, bitwise."""
    g = gold("picp_test1009.npz")
    ocam = OCam(g["rows"], g["cols"], g["z_near"], g["z_far"], g["K"], np.eye(4))
    r = o32.picp_solve_raw(ocam, g["world"], g["cur_pts"], g["corr"], 100, 10000.0, False)
    H, b, st, T = exact_trace(vo, ctx, g, g["world"], g["cur_pts"], g["corr"], 100, 10000.0, False)
    assert np.array_equal(T, r["T"]) and np.array_equal(H, r["H"]) and np.array_equal(b, r["b"]) and np.array_equal(st, r["stats"])
    s = vo.PICPSolver(ctx)
    s.setExact(True)
    s.setKernelThreshold(10000.0)
    s.init(vo.Camera(g["rows"], g["cols"], g["z_near"], g["z_far"], g["K"], np.eye(4), ctx=ctx), g["world"], g["cur_pts"])
    s.solve(g["corr"], False, 100)
    assert np.array_equal(s.camera().worldInCameraPose(), r["T"][-1])
    assert np.array_equal(s.system()[0], r["H"][-1]) and s.numInliers() == int(r["stats"][-1, 2])
    # back to the fast mode on the same handle: rounding-level agreement only
    s.setExact(False)
    s.init(vo.Camera(g["rows"], g["cols"], g["z_near"], g["z_far"], g["K"], np.eye(4), ctx=ctx), g["world"], g["cur_pts"])
    s.solve(g["corr"], False, 100)
    assert np.abs(s.camera().worldInCameraPose() - r["T"][-1]).max() < 1e-4
    s.close()


def test_exact_many_chunks_and_dropped_terms(vo, ctx, o32):
    """more correspondences than one staging pass (256), pairs whose projection is gated out mixed in, a start pose
    away from the identity"""
    fp = vo.synth.frame_pair(3000, seed=91, drop=0.05, distractors=30, model_drop=0.05)
    m = vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=ctx)
    j = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
    rng = np.random.default_rng(5)
    world = fp["model"].copy()
    far = rng.choice(len(world), 200, replace=False)
    world[far, 2] += np.float32(40.0)                 # beyond z_far: skipped before any statistic
    T0 = vo.synth.random_isometry(rng, 0.02, 0.05)
    g = dict(rows=fp["rows"], cols=fp["cols"], z_near=fp["z_near"], z_far=fp["z_far"], K=fp["K"])
    for thr, keep in ((10000.0, False), (30.0, True)):
        r = o32.picp_solve_raw(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], T0), world, fp["cur_pts"], j,
                               8, thr, keep)
        H, b, st, T = exact_trace(vo, ctx, g, world, fp["cur_pts"], j, 8, thr, keep, T0=T0)
        assert len(j) > 2000 and st[0, 2] < len(j)
        assert np.array_equal(H, r["H"]) and np.array_equal(b, r["b"]) and np.array_equal(st, r["stats"]) and np.array_equal(T, r["T"])


def test_exact_batched_form(vo, ctx, o32):
    """vo_picp_batch_set_form(3): the batched solver in reference-order arithmetic, one workgroup per problem"""
    P, n, n_it = 5, 700, 9
    fps = [vo.synth.frame_pair(n, seed=300 + p, drop=0.0, distractors=0, model_drop=0.0) for p in range(P)]
    nm = max(len(f["model"]) for f in fps); nz = max(len(f["cur_pts"]) for f in fps)
    joins = []
    for f in fps:
        m = o32.match(f["ref_app"], f["cur_app"])
        joins.append(o32.join(m, f["model_pairs"]))
    np_max = max(len(j) for j in joins)
    world = np.zeros((P, nm, 3), np.float32); meas = np.zeros((P, nz, 2), np.float32)
    pairs = np.zeros((P, np_max, 2), np.int32); cnt = np.zeros(P, np.int32)
    for p, f in enumerate(fps):
        world[p, :len(f["model"])] = f["model"]; meas[p, :len(f["cur_pts"])] = f["cur_pts"]
        k = len(joins[p]) - 13 * p                    # ragged counts
        pairs[p, :k] = joins[p][:k]; cnt[p] = k
    f0 = fps[0]
    lib = ctx.lib
    dw, dm, dp, dn = ctx.to_device(world), ctx.to_device(meas), ctx.to_device(pairs), ctx.to_device(cnt)
    dT, ds = ctx.alloc(P * 64), ctx.alloc(P * 16)
    Kc = np.ascontiguousarray(np.asarray(f0["K"], np.float32).T).ravel()
    try:
        assert lib.vo_picp_batch_set_form(ctx.h, 3) == 0
        rc = lib.vo_picp_solve_batch_dev(ctx.h, P, f0["rows"], f0["cols"], f0["z_near"], f0["z_far"],
                                         Kc.ctypes.data_as(C.c_void_p), C.c_float(10000.0), 0, C.c_void_p(dw), C.c_size_t(nm),
                                         C.c_void_p(dm), C.c_size_t(nz), C.c_void_p(dp), C.c_size_t(np_max), C.c_void_p(dn),
                                         C.c_void_p(0), n_it, C.c_void_p(dT), C.c_void_p(ds))
        assert rc == 0, lib.vo_last_error()
        T = np.zeros((P, 16), np.float32); st = np.zeros((P, 4), np.float32)
        ctx.d2h(T, dT); ctx.d2h(st, ds)
    finally:
        lib.vo_picp_batch_set_form(ctx.h, 0)
        for d in (dw, dm, dp, dn, dT, ds):
            ctx.free(d)
    for p, f in enumerate(fps):
        r = o32.picp_solve_raw(OCam(f["rows"], f["cols"], f["z_near"], f["z_far"], f["K"], np.eye(4)), world[p], meas[p],
                               pairs[p, :cnt[p]], n_it, 10000.0, False)
        assert np.array_equal(T[p].reshape(4, 4).T, r["T"][-1])
        assert np.array_equal(st[p, :3], r["stats"][-1])


def test_one_round_honours_in_place_edits(vo, ctx, o32):
    """The reference reads the correspondence vector on every oneRound (picp_solver.cpp:62).  Editing the caller's
    array in place between two rounds (same address, same size) must be seen: prune outliers after the first round."""
    fp = vo.synth.frame_pair(4000, seed=17, drop=0.0, distractors=0, model_drop=0.0)
    m = o32.match(fp["ref_app"], fp["cur_app"])
    j = np.ascontiguousarray(o32.join(m, fp["model_pairs"]).astype(np.int32))
    assert len(j) > 3000
    ocam = OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4))
    for exact in (True, False):
        jj = j.copy()
        s = vo.PICPSolver(ctx)
        s.setExact(exact)
        s.setKernelThreshold(10000.0)
        s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
        s.oneRound(jj, False)
        T1 = s.camera().worldInCameraPose()
        r1 = o32.picp_solve(ocam, fp["model"], fp["cur_pts"], jj, 1, 10000.0, False, trace=False)
        addr = jj.ctypes.data
        # in place: re-point a handful of pairs in the middle of the array at wrong model points (none of the
        # positions a strided 64-sample fingerprint would look at: 1..62 apart from multiples of len/64)
        step = len(jj) // 64
        idx = [k for k in range(step + 1, step + 40) if k % step != 0 and k != len(jj) - 1][:25]
        jj[idx, 1] = jj[[k + 500 for k in idx], 1]
        assert jj.ctypes.data == addr
        s.oneRound(jj, False)
        T2 = s.camera().worldInCameraPose()
        n2 = s.numInliers()
        r2 = o32.picp_solve(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], r1["T"]), fp["model"], fp["cur_pts"],
                            jj, 1, 10000.0, False, trace=False)
        stale = o32.picp_solve(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], r1["T"]), fp["model"],
                               fp["cur_pts"], j, 1, 10000.0, False, trace=False)
        # the edit matters: the re-pointed pairs turn into outliers, so a solver that kept the old pairs is found out
        assert stale["num_inliers"] != r2["num_inliers"] and not np.array_equal(stale["T"], r2["T"])
        if exact:
            assert np.array_equal(T1, r1["T"]) and np.array_equal(T2, r2["T"]) and n2 == r2["num_inliers"]
        else:
            assert np.abs(T2 - r2["T"]).max() < 1e-4 and n2 == r2["num_inliers"]
        # the explicit pair: set once, iterate without any comparison
        s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
        s.setCorrespondences(jj)
        s.rounds(False, 1)
        s.rounds(False, 1)
        r3 = o32.picp_solve(ocam, fp["model"], fp["cur_pts"], jj, 2, 10000.0, False, trace=False)
        if exact:
            assert np.array_equal(s.camera().worldInCameraPose(), r3["T"])
        else:
            assert np.abs(s.camera().worldInCameraPose() - r3["T"]).max() < 1e-4
        s.close()


def test_rounds_after_a_solve_with_other_pairs_continue_on_those_pairs(vo, ctx, o32):
    """setCorrespondences(A) ; solve(B) ; init (new points: the packing is invalidated) ; rounds() used to
    re-pack len(A) pairs out of a buffer that held B followed by the tail of A.  Now the hand-over count follows the
    array the solver last received: rounds() continues on B -- bit for bit in reference-order mode."""
    fp = vo.synth.frame_pair(3000, seed=23, drop=0.0, distractors=0, model_drop=0.0)
    m = o32.match(fp["ref_app"], fp["cur_app"])
    A = np.ascontiguousarray(o32.join(m, fp["model_pairs"]).astype(np.int32))
    B = np.ascontiguousarray(A[100:1300][::-1])           # fewer pairs, other order
    assert len(A) > 2500
    ocam = OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4))
    cam = vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx)
    for exact in (True, False):
        s = vo.PICPSolver(ctx)
        s.setExact(exact)
        s.setKernelThreshold(10000.0)
        s.init(cam, fp["model"], fp["cur_pts"])
        s.setCorrespondences(A)
        s.rounds(False, 1)
        s.solve(B, False, 2)                              # another array through the comparing entry point
        s.init(cam, fp["model"], fp["cur_pts"])           # pose back to the identity, packing invalidated
        s.rounds(False, 3)
        want = o32.picp_solve(ocam, fp["model"], fp["cur_pts"], B, 3, 10000.0, False, trace=False)
        got = s.camera().worldInCameraPose()
        assert s.numInliers() == want["num_inliers"] == len(B)
        if exact:
            assert np.array_equal(got, want["T"])
        else:
            assert np.abs(got - want["T"]).max() < 1e-4
        s.close()


def test_graph_bookkeeping_is_visible(vo, ctx):
    """a multi-round solve replays a captured graph; the handle says so (vo_picp_graph_info), and a failed capture
    would be counted and reported through vo_last_error() instead of silently dropping to plain launches"""
    fp = vo.synth.frame_pair(2000, seed=29, drop=0.0, distractors=0, model_drop=0.0)
    s = vo.PICPSolver(ctx)
    s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
    j = np.stack([np.arange(1500), np.arange(1500)], 1).astype(np.int32)
    s.solve(j, False, 5)
    s.solve(j, False, 5)
    s.solve(j, False, 7)
    use, n_graphs, n_fail = s.graphInfo()
    assert (use, n_graphs, n_fail) == (1, 2, 0)
    s.close()


def _run_app(tmp_path, *flags):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s"])
    r = subprocess.run([os.path.join(BIN, "vo_complete"), DATA, str(tmp_path), *flags], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    poses = np.loadtxt(os.path.join(tmp_path, "poses_raw.txt"), dtype=np.float64).astype(np.float32).reshape(-1, 4, 4)
    return r.stdout, poses


def test_example_data_chain_is_bit_identical_to_ref32(tmp_path, o32):
    """BASELINE configs[4]: apps/vo_complete --exact on the reference's own dataset against the oracle-side run of the
    same loop started from the same first relative pose: all 119 frames' (matches, joined pairs, INLIERS) equal and all
    121 camera poses equal bit for bit."""
    out, poses = _run_app(tmp_path, "--exact")
    assert poses.shape == (121, 4, 4)
    res = vp.run_vo_complete(DATA, rounds=100, o=o32, X0=poses[1])
    counts = re.findall(r"^meas-\d+\.dat: (\d+) matches, (\d+) model correspondences, (\d+) inliers", out, flags=re.M)
    assert len(counts) == 119
    assert np.array_equal(np.array(counts, dtype=int), np.array(res["stats"], dtype=int))
    ref = np.array(res["trajectory"], dtype=np.float32)
    assert np.array_equal(poses, ref)
    # the map the loop body keeps (vo_complete.cpp:145-147,175-176,183): map_raw.txt = map.txt + map_appearances.txt at full
    # precision, every entry in order, bit for bit -- host upkeep (PointCloudVector::update behind the facade) here ...
    want_map = np.concatenate([res["map"], res["map_app"]], axis=1).astype(np.float32)

    def map_of(d):
        return np.loadtxt(os.path.join(d, "map_raw.txt"), dtype=np.float64).astype(np.float32).reshape(-1, 13)
    assert len(want_map) > 400 and np.array_equal(map_of(tmp_path), want_map)
    # the epipolar initialisation itself (f-1): host double Jacobi vs the oracle's numpy SVDs
    res0 = vp.run_vo_complete(DATA, rounds=1, o=o32)
    assert np.abs(poses[1] - res0["trajectory"][1]).max() < 2e-5
    # and the device-resident form of the loop runs the same chain
    rdir = tmp_path / "resident"
    rdir.mkdir()
    _, poses_r = _run_app(rdir, "--exact", "--resident")
    assert np.array_equal(poses_r, poses)
    assert np.array_equal(map_of(rdir), want_map)            # ... and the device map inside the chain (vo_map_*) there
    # ... also with all 120 consecutive pairs of the dataset (14..127 points per frame) matched by ONE batched call up front
    udir = tmp_path / "upfront"
    udir.mkdir()
    _, poses_u = _run_app(udir, "--exact", "--resident", "--match-up-front")
    assert np.array_equal(poses_u, poses)
    assert np.array_equal(map_of(udir), want_map)
