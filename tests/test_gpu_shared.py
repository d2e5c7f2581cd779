"""Batched solver with fewer problems than CUs (csrc/picp.hip, picp_batch_shared_kernel): the launch has one workgroup
per CU, and the waves of the workgroups without a problem take chunks of the others' correspondences every round.  The
sums a helper wave hands over must be exactly what the problem's own workgroup would have computed in its place: the result
may depend on the problem count and sizes of the call (they fix the chunks), never on timing."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.oracle import Camera as OCam

pytestmark = pytest.mark.gpu
ENV = ("VO_PICP_SHARE", "VO_PICP_HELP_KEEP", "VO_PICP_HELP_G", "VO_PICP_HELP_SLACK", "VO_PICP_HELP_ABSENT")


@pytest.fixture(autouse=True)
def _clean_env():
    saved = {k: os.environ.pop(k, None) for k in ENV}
    yield
    for k, v in saved.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


def _corr(fp):
    mp = dict(fp["model_pairs"].tolist())
    return np.array([(c, mp[r]) for r, c in fp["gt_matches"].tolist()], np.int32)


class Batch:
    """P problems over ONE generated frame pair (so that the oracle has one camera and the test one upload): problem p uses
    the first sizes[p] pairs and starts at T0[p]"""

    def __init__(self, vo, ctx, n, sizes, seed, K=None, rng_seed=3):
        self.vo, self.ctx, self.n = vo, ctx, n
        self.fp = vo.synth.frame_pair(n, seed=seed, distractors=n // 50)
        self.pairs = _corr(self.fp)
        self.sizes = np.array([min(s, len(self.pairs)) for s in sizes], np.int32)
        self.P = len(sizes)
        self.stride = len(self.pairs)
        self.K = np.asarray(self.fp["K"] if K is None else K, np.float32)
        rng = np.random.default_rng(rng_seed)
        self.T0 = np.stack([vo.synth.random_isometry(rng, 0.01, 0.02) for _ in range(self.P)]).astype(np.float32)
        self.d = [ctx.to_device(np.tile(self.fp["model"], (self.P, 1))), ctx.to_device(np.tile(self.fp["cur_pts"], (self.P, 1))),
                  ctx.to_device(np.tile(self.pairs, (self.P, 1))), ctx.to_device(self.sizes),
                  ctx.to_device(np.ascontiguousarray(np.transpose(self.T0, (0, 2, 1))).reshape(self.P, 16))]
        self.d_T, self.d_S = ctx.alloc(self.P * 64), ctx.alloc(self.P * 16)

    def run(self, iters, thr, keep, form=2):
        lib, ctx = self.ctx.lib, self.ctx
        assert lib.vo_picp_batch_set_form(ctx.h, form) == 0
        K = np.ascontiguousarray(self.K.T).ravel()
        n_pts = len(self.fp["model"])
        rc = lib.vo_picp_solve_batch_dev(ctx.h, self.P, 480, 640, 0, 10, K.ctypes.data_as(C.c_void_p), C.c_float(thr), int(keep),
                                         C.c_void_p(self.d[0]), C.c_size_t(n_pts), C.c_void_p(self.d[1]), C.c_size_t(len(self.fp["cur_pts"])),
                                         C.c_void_p(self.d[2]), C.c_size_t(self.stride), C.c_void_p(self.d[3]), C.c_void_p(self.d[4]),
                                         iters, C.c_void_p(self.d_T), C.c_void_p(self.d_S))
        assert rc == 0, lib.vo_last_error()
        f, w = C.c_int(), C.c_int()
        assert lib.vo_picp_batch_info(ctx.h, C.byref(f), C.byref(w)) == 0
        T = np.zeros((self.P, 16), np.float32); S = np.zeros((self.P, 4), np.float32)
        ctx.d2h(T, self.d_T); ctx.d2h(S, self.d_S)
        lib.vo_picp_batch_set_form(ctx.h, 0)
        return T, S, f.value, w.value

    def close(self):
        for x in self.d + [self.d_T, self.d_S]:
            self.ctx.free(x)


@pytest.mark.parametrize("keep,pinhole", [(False, True), (True, True), (False, False), (True, False)])
def test_shared_form_against_the_oracle_on_ragged_problems(vo, ctx, o32, keep, pinhole):
    n, iters, thr = 24000, 5, 60.0
    # empty, a few, below / at / above the trips a workgroup keeps, everything; the last three repeat the sizes of earlier ones
    sizes = [0, 3, 2000, 6143, 6144, 6148, 9217, 12288, 15000, 18431, 18432, 18435, 20000, 22001, 10 ** 9, 10 ** 9, 12288, 3, 0, 22001]
    b = Batch(vo, ctx, n, sizes, seed=7100 + int(keep))
    if not pinhole:
        K = b.K.copy(); K[0, 1] = 0.7                          # a skew term: the general projection
        b.close()
        b = Batch(vo, ctx, n, sizes, seed=7100 + int(keep), K=K)
    for p, q in ((16, 7), (17, 1), (18, 0), (19, 13), (15, 14)):
        b.T0[p] = b.T0[q]
    ctx.free(b.d[4]); b.d[4] = ctx.to_device(np.ascontiguousarray(np.transpose(b.T0, (0, 2, 1))).reshape(b.P, 16))
    T, S, form, wgs = b.run(iters, thr, keep)
    assert form == 4 and wgs > b.P                           # the launch had helpers
    for p in range(b.P):
        r = o32.picp_solve(OCam(480, 640, 0, 10, b.K, b.T0[p]), b.fp["model"], b.fp["cur_pts"], b.pairs[: b.sizes[p]], iters, thr, keep, trace=False)
        assert np.abs(T[p].reshape(4, 4).T - r["T"]).max() < 1e-4, p
        assert abs(int(S[p, 2]) - r["num_inliers"]) <= 1, p
        assert abs(S[p, 0] - r["chi_inliers"]) <= 2e-4 * max(1.0, r["chi_inliers"]), p
    for p, q in ((16, 7), (17, 1), (18, 0), (19, 13), (15, 14)):   # same data in one call: same bits, wherever the problem sits
        assert T[p].tobytes() == T[q].tobytes() and S[p].tobytes() == S[q].tobytes(), (p, q)
    assert np.array_equal(T[0].reshape(4, 4).T, b.T0[0])     # no correspondence: H = I, b = 0, the pose stays
    # zero rounds: the starting poses come back
    T0r, S0r, form0, _ = b.run(0, thr, keep)
    assert np.array_equal(T0r.reshape(-1, 4, 4).transpose(0, 2, 1), b.T0) and not S0r.any()
    b.close()


def test_result_does_not_depend_on_the_helpers_timing_or_presence(vo, ctx):
    n, iters, thr = 30000, 12, 10000.0
    rng = np.random.default_rng(11)
    for sizes in ([10 ** 9] * 24, list(rng.integers(0, 31000, 40)), [10 ** 9] * 3, [10 ** 9] * 150):
        b = Batch(vo, ctx, n, sizes, seed=7200)
        ref = b.run(iters, thr, False)
        assert ref[2] == 4
        for _ in range(3):                                   # run to run
            again = b.run(iters, thr, False)
            assert again[0].tobytes() == ref[0].tobytes() and again[1].tobytes() == ref[1].tobytes()
        os.environ["VO_PICP_HELP_ABSENT"] = "1"              # no helper wave ever delivers: every home stands in for all its chunks
        alone = b.run(iters, thr, False)
        os.environ.pop("VO_PICP_HELP_ABSENT")
        assert alone[2] == 4 and alone[0].tobytes() == ref[0].tobytes() and alone[1].tobytes() == ref[1].tobytes()
        # other chunks (another keep, another chunk length), no helpers at all, one launch per round: other summation orders
        os.environ["VO_PICP_HELP_KEEP"] = "3"; os.environ["VO_PICP_HELP_G"] = "2"
        other = b.run(iters, thr, False)
        os.environ.pop("VO_PICP_HELP_KEEP"); os.environ.pop("VO_PICP_HELP_G")
        os.environ["VO_PICP_SHARE"] = "0"
        plain = b.run(iters, thr, False)
        os.environ.pop("VO_PICP_SHARE")
        assert other[2] == 4 and plain[2] == 2
        rounds = b.run(iters, thr, False, form=1)
        assert rounds[2] == 1
        for x in (other, plain, rounds):
            assert np.abs(x[0] - ref[0]).max() < 2e-5 and np.abs(x[1][:, 2] - ref[1][:, 2]).max() <= 1
        b.close()


def test_rule_which_calls_get_helpers(vo, ctx):
    """auto mode (form 0): a few problems -> one launch per round; up to 0.65 problems per CU -> helpers; more, or short
    problems -> one workgroup per problem alone"""
    n_cu = 256
    for P, n, want in ((2, 30000, 1), (16, 30000, 4), (64, 30000, 4), (int(0.65 * n_cu), 20000, 4), (int(0.65 * n_cu) + 8, 20000, 2), (64, 9000, 2)):
        b = Batch(vo, ctx, n, [10 ** 9] * P, seed=7300)
        T, S, form, wgs = b.run(3, 10000.0, False, form=0)
        assert form == want, (P, n, form)
        if form == 4:
            assert wgs == n_cu
        assert np.isfinite(T).all() and (S[:, 2] > 0.9 * min(n, len(b.pairs))).all()
        b.close()


def test_frames_call_with_helpers_in_its_solver_stage(vo, ctx, o32):
    """vo_frames_batch_dev over 40 frame pairs x 20 000 points (0.16 problems per CU): the solver stage runs with helper
    waves; every frame against the generator, copies of one pair identical, three frames stage by stage against the oracle"""
    N, F, ITERS = 20000, 40, 20
    distinct = [vo.synth.frame_pair(N, seed=7400 + p) for p in range(5)]
    fps = [distinct[i % 5] for i in range(F)]
    bp = vo.BatchPipeline(ctx, lambda lo, hi: fps[lo:hi], n_iters=ITERS, n_frames=F, upload_block=20)
    bp.run()
    ctx.synchronize()
    f_, w_ = C.c_int(), C.c_int()
    assert ctx.lib.vo_picp_batch_info(ctx.h, C.byref(f_), C.byref(w_)) == 0 and f_.value == 4 and w_.value > F
    c, T, st = bp.counts(), bp.poses(), bp.stats()
    assert np.all(c[0] == N) and np.all(c[1] == N) and np.all(st[:, 2] == N)
    assert np.abs(T - bp.X_gt).max() < 1e-3
    for f in range(F):
        assert np.array_equal(T[f], T[f % 5]) and np.array_equal(st[f], st[f % 5]) and np.array_equal(c[:, f], c[:, f % 5])
    for f in (0, 17, 39):
        fp = fps[f]
        j_o = o32.join(o32.match_kdtree(fp["ref_app"], fp["cur_app"]), fp["model_pairs"], linear=True)
        assert np.array_equal(bp.fetch("join", f), j_o)
        r = o32.picp_solve(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)), fp["model"],
                           fp["cur_pts"], j_o, ITERS, 10000.0, False, trace=False)
        assert np.abs(T[f] - r["T"]).max() < 1e-4 and int(st[f, 2]) == r["num_inliers"]
        xo, po, ao = o32.triangulate(fp["K"], T[f], o32.match_kdtree(fp["ref_app"], fp["cur_app"]), fp["ref_pts"], fp["cur_pts"], fp["cur_app"])
        assert np.array_equal(bp.fetch("tri_pairs", f), po) and np.array_equal(bp.fetch("tri_xyz", f), xo)
    bp.close()
