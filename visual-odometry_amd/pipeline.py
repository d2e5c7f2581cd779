"""Device-resident frame step of the VO loop (vo_complete.cpp:150-179):

    match -> join -> rigid transform of the model -> n x oneRound -> triangulate

Every stage is enqueued on the context's stream through the *_dev entry points
of the C ABI; counts travel between stages in device memory, so a frame costs
no host synchronisation.  Used by bench.py and the GPU tests; the production
host loop is the C++ one in apps/.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .api import Context, Event, _chk, _colmajor, _ptr


class FramePipeline:
    def __init__(self, ctx: Context, fp: dict, n_iters: int = 50, kernel_threshold: float = 10000.0):
        """fp: a synth.frame_pair() dict; its arrays are uploaded once (inputs
        resident in HBM before any timed region starts)."""
        self.ctx, self.lib = ctx, ctx.lib
        self.n_iters = n_iters
        self.K = _colmajor(fp["K"], 3)
        self.cam = (int(fp["rows"]), int(fp["cols"]), int(fp["z_near"]), int(fp["z_far"]))
        self.n_ref, self.n_cur = len(fp["ref_app"]), len(fp["cur_app"])
        self.n_model, self.n_mp = len(fp["model"]), len(fp["model_pairs"])
        self.nq = min(self.n_ref, self.n_cur)
        up = ctx.to_device
        self.d_ref_app, self.d_cur_app = up(fp["ref_app"]), up(fp["cur_app"])
        self.d_ref_pts, self.d_cur_pts = up(fp["ref_pts"]), up(fp["cur_pts"])
        self.d_model, self.d_model_pairs = up(fp["model"]), up(fp["model_pairs"])
        a = ctx.alloc
        self.d_m = a(max(self.nq, 1) * 8)          # (ref_idx, cur_idx)
        self.d_j = a(max(self.nq, 1) * 8)          # (cur_idx, model_idx)
        self.d_model_t = a(max(self.n_model, 1) * 12)
        self.d_tri_xyz = a(max(self.nq, 1) * 12)
        self.d_tri_pairs = a(max(self.nq, 1) * 8)
        self.d_tri_app = a(max(self.nq, 1) * 40)
        self.d_counts = a(64)                       # [0]=n_match [1]=n_join [2]=n_tri
        self.d_ident = up(np.eye(4, dtype=np.float32))
        self.X_prev = _colmajor(np.eye(4), 4)       # pose of the previous frame (vo_complete.cpp:159)
        h = C.c_void_p()
        _chk(self.lib.vo_picp_create(ctx.h, C.byref(h)))
        self.solver = h
        _chk(self.lib.vo_picp_set_camera(h, *map(C.c_int, self.cam), _ptr(self.K), _ptr(_colmajor(np.eye(4), 4))))
        _chk(self.lib.vo_picp_set_kernel_threshold(h, C.c_float(kernel_threshold)))
        _chk(self.lib.vo_picp_set_points_dev(h, C.c_void_p(self.d_model_t), C.c_int(self.n_model),
                                             C.c_void_p(self.d_cur_pts), C.c_int(self.n_cur)))
        p = C.c_void_p()
        _chk(self.lib.vo_picp_pose_dev_ptr(h, C.byref(p)))
        self.d_pose = p.value                       # the solver's own 4x4, read in place by triangulate

    def _cnt(self, i):
        return C.c_void_p(self.d_counts + 4 * i)

    # -- stages (all asynchronous) -------------------------------------------
    def match(self):
        _chk(self.lib.vo_match_appearances_dev(self.ctx.h, C.c_void_p(self.d_ref_app), C.c_int(self.n_ref),
                                               C.c_void_p(self.d_cur_app), C.c_int(self.n_cur), C.c_float(0.1),
                                               C.c_void_p(self.d_m), self._cnt(0)))

    def join(self):
        _chk(self.lib.vo_join_correspondences_dev(self.ctx.h, C.c_void_p(self.d_m), C.c_int(self.nq), self._cnt(0),
                                                  C.c_void_p(self.d_model_pairs), C.c_int(self.n_mp), None,
                                                  C.c_int(self.n_ref), C.c_void_p(self.d_j), self._cnt(1)))

    def transform(self):
        _chk(self.lib.vo_transform_points_dev(self.ctx.h, _ptr(self.X_prev), None, C.c_void_p(self.d_model),
                                              C.c_int(self.n_model), None, C.c_void_p(self.d_model_t)))

    def picp(self):
        """solver re-initialised at identity every frame (vo_complete.cpp:161-164)"""
        _chk(self.lib.vo_picp_set_pose_dev(self.solver, C.c_void_p(self.d_ident)))
        _chk(self.lib.vo_picp_solve_dev(self.solver, C.c_void_p(self.d_j), C.c_int(self.nq), self._cnt(1),
                                        C.c_int(0), C.c_int(self.n_iters)))

    def triangulate(self):
        _chk(self.lib.vo_triangulate_dev(self.ctx.h, _ptr(self.K), None, C.c_void_p(self.d_pose),
                                         C.c_void_p(self.d_m), C.c_int(self.nq), self._cnt(0),
                                         C.c_void_p(self.d_ref_pts), C.c_int(self.n_ref),
                                         C.c_void_p(self.d_cur_pts), C.c_int(self.n_cur),
                                         C.c_void_p(self.d_cur_app), C.c_void_p(self.d_tri_xyz),
                                         C.c_void_p(self.d_tri_pairs), C.c_void_p(self.d_tri_app), self._cnt(2)))

    def frame(self):
        self.match(); self.join(); self.transform(); self.picp(); self.triangulate()

    def capture_frame(self):
        """Capture one whole frame into a hipGraph (after at least one plain frame(), which sizes
        every internal buffer); frame_graph() then replays it with a single launch."""
        self.frame()
        self.ctx.synchronize()
        g = C.c_void_p()
        _chk(self.lib.vo_ctx_begin_capture(self.ctx.h))
        try:
            self.frame()
        finally:
            _chk(self.lib.vo_ctx_end_capture(self.ctx.h, C.byref(g)))
        self.graph = g

    def frame_graph(self):
        _chk(self.lib.vo_graph_launch(self.graph))

    # -- results (synchronise) -------------------------------------------------
    def counts(self):
        c = np.zeros(3, np.int32)
        self.ctx.d2h(c, self.d_counts)
        return c

    def pose(self):
        T = np.zeros(16, np.float32)
        self.ctx.d2h(T, self.d_pose)
        return T.reshape(4, 4).T.copy()

    def fetch(self, what):
        c = self.counts()
        spec = {"match": (self.d_m, c[0], 2, np.int32), "join": (self.d_j, c[1], 2, np.int32),
                "tri_xyz": (self.d_tri_xyz, c[2], 3, np.float32), "tri_pairs": (self.d_tri_pairs, c[2], 2, np.int32),
                "tri_app": (self.d_tri_app, c[2], 10, np.float32)}[what]
        out = np.zeros((int(spec[1]), spec[2]), spec[3])
        if len(out):
            self.ctx.d2h(out, spec[0])
        return out

    def stats(self):
        ci, co, ni = C.c_float(), C.c_float(), C.c_int()
        _chk(self.lib.vo_picp_get_stats(self.solver, C.byref(ci), C.byref(co), C.byref(ni)))
        return ci.value, co.value, ni.value

    def close(self):
        if getattr(self, "graph", None):
            self.lib.vo_graph_destroy(self.graph)
            self.graph = None
        if self.solver:
            self.lib.vo_picp_destroy(self.solver)
            self.solver = None
        for d in (self.d_ref_app, self.d_cur_app, self.d_ref_pts, self.d_cur_pts, self.d_model, self.d_model_pairs,
                  self.d_m, self.d_j, self.d_model_t, self.d_tri_xyz, self.d_tri_pairs, self.d_tri_app,
                  self.d_counts, self.d_ident):
            self.ctx.free(d)


class _FrameBatch(C.Structure):
    """vo_frame_batch of include/vo_hip.h"""
    _fields_ = [("n_frames", C.c_int), ("n_ref", C.c_int), ("n_cur", C.c_int), ("n_model", C.c_int),
                ("n_model_pairs", C.c_int),
                ("ref_app", C.c_void_p), ("cur_app", C.c_void_p), ("ref_pts", C.c_void_p), ("cur_pts", C.c_void_p),
                ("model", C.c_void_p), ("model_pairs", C.c_void_p), ("X_prev", C.c_void_p),
                ("rows", C.c_int), ("cols", C.c_int), ("z_near", C.c_int), ("z_far", C.c_int),
                ("K", C.c_float * 9), ("kernel_threshold", C.c_float), ("keep_outliers", C.c_int),
                ("n_iters", C.c_int), ("radius", C.c_float),
                ("matches", C.c_void_p), ("joined", C.c_void_p), ("model_moved", C.c_void_p), ("poses", C.c_void_p),
                ("stats", C.c_void_p), ("tri_xyz", C.c_void_p), ("tri_pairs", C.c_void_p), ("tri_app", C.c_void_p),
                ("counts", C.c_void_p)]


class BatchPipeline:
    """n_frames independent frame pairs (vo_frames_batch_dev): every stage is one batched launch per call, the solver
    is the batched kernel.  All frames must have the same set sizes.  `fps` is a list of frame-pair dicts, or -- for
    many frames -- a callable gen(lo, hi) -> list of the dicts of frames lo..hi-1, asked for `upload_block` frames at a
    time so that the host never holds the whole set; `frames_per_call` splits run() into several calls."""

    _IN = (("ref_app", np.float32, 10), ("cur_app", np.float32, 10), ("ref_pts", np.float32, 2), ("cur_pts", np.float32, 2),
           ("model", np.float32, 3), ("model_pairs", np.int32, 2))

    def __init__(self, ctx: Context, fps, n_iters: int = 50, kernel_threshold: float = 10000.0,
                 with_appearance: bool = True, poses_ptr: int | None = None, n_frames: int | None = None,
                 upload_block: int = 100, frames_per_call: int | None = None, with_moved: bool = False, X_prev=None):
        """poses_ptr: optional device buffer (n_frames*16 floats, e.g. a torch tensor's data_ptr) that
        receives the poses directly, so that a collective can read them without a copy.
        with_moved: also produce X_prev * model as an output array (vo_frame_batch.model_moved); without it the solver's
        gather moves the points it fetches itself.  X_prev: optional (F, 4, 4) poses of the previous frames
        (vo_complete.cpp:159); default: identity."""
        self.ctx, self.lib = ctx, ctx.lib
        gen = fps if callable(fps) else (lambda lo, hi: fps[lo:hi])
        F = self.F = int(n_frames if n_frames is not None else len(fps))
        first = gen(0, min(upload_block, F))
        f0 = first[0]
        self.n_ref, self.n_cur = len(f0["ref_app"]), len(f0["cur_app"])
        self.n_model, self.n_mp = len(f0["model"]), len(f0["model_pairs"])
        per = {"ref_app": self.n_ref, "cur_app": self.n_cur, "ref_pts": self.n_ref, "cur_pts": self.n_cur,
               "model": self.n_model, "model_pairs": self.n_mp}
        self.q = min(self.n_ref, self.n_cur)
        a = ctx.alloc
        self._in = [a(max(F * per[k] * w * 4, 16)) for k, _, w in self._IN]
        self.X_gt = np.zeros((F, 4, 4), np.float32)      # ground truth of the generator, when the dicts carry it
        lo = 0
        while lo < F:
            blk = first if lo == 0 else gen(lo, min(lo + upload_block, F))
            for f in blk:
                assert (len(f["ref_app"]), len(f["cur_app"]), len(f["model"]), len(f["model_pairs"])) == \
                    (self.n_ref, self.n_cur, self.n_model, self.n_mp), "frames of a batch must have identical sizes"
            for d, (k, dt, w) in zip(self._in, self._IN):
                arr = np.ascontiguousarray(np.stack([np.asarray(f[k], dt).reshape(per[k], w) for f in blk]))
                ctx.h2d(d + lo * per[k] * w * 4, arr)
            for i, f in enumerate(blk):
                if "X_gt" in f:
                    self.X_gt[lo + i] = f["X_gt"]
            lo += len(blk)
            del blk
        q = self.q
        self.d_matches, self.d_joined = a(F * q * 8), a(F * q * 8)
        self._own_poses = poses_ptr is None
        self.d_moved = a(F * self.n_model * 12) if with_moved else 0
        self.d_X = ctx.to_device(np.ascontiguousarray(np.stack([_colmajor(X, 4) for X in X_prev]))) if X_prev is not None else 0
        self.d_poses, self.d_stats = (a(F * 64) if poses_ptr is None else poses_ptr), a(F * 16)
        self.d_tri_xyz, self.d_tri_pairs = a(F * q * 12), a(F * q * 8)
        self.d_tri_app = a(F * q * 40) if with_appearance else 0
        self.d_counts = a(3 * F * 4)
        per_call = F if not frames_per_call else max(1, min(int(frames_per_call), F))
        self.calls = []
        for lo in range(0, F, per_call):
            hi = min(lo + per_call, F)
            b = _FrameBatch()
            b.n_frames, b.n_ref, b.n_cur, b.n_model, b.n_model_pairs = hi - lo, self.n_ref, self.n_cur, self.n_model, self.n_mp
            (b.ref_app, b.cur_app, b.ref_pts, b.cur_pts, b.model, b.model_pairs) = \
                [d + lo * per[k] * w * 4 for d, (k, _, w) in zip(self._in, self._IN)]
            b.X_prev = (self.d_X + lo * 64) if self.d_X else None
            b.rows, b.cols, b.z_near, b.z_far = int(f0["rows"]), int(f0["cols"]), int(f0["z_near"]), int(f0["z_far"])
            b.K[:] = _colmajor(f0["K"], 3).tolist()
            b.kernel_threshold, b.keep_outliers, b.n_iters, b.radius = kernel_threshold, 0, n_iters, 0.1
            b.matches, b.joined = self.d_matches + lo * q * 8, self.d_joined + lo * q * 8
            b.model_moved = (self.d_moved + lo * self.n_model * 12) if self.d_moved else None
            b.poses, b.stats = self.d_poses + lo * 64, self.d_stats + lo * 16
            b.tri_xyz, b.tri_pairs = self.d_tri_xyz + lo * q * 12, self.d_tri_pairs + lo * q * 8
            b.tri_app = (self.d_tri_app + lo * q * 40) if self.d_tri_app else None
            b.counts = self.d_counts + 3 * lo * 4        # this call's [3][hi - lo] block
            self.calls.append((lo, hi, b))
        self.b = self.calls[0][2]

    def run(self):
        for _, _, b in self.calls:
            _chk(self.lib.vo_frames_batch_dev(self.ctx.h, C.byref(b)))

    def match_only(self):
        """the matcher stage of every call alone (vo_match_appearances_batch_dev on the same inputs, into the same outputs)"""
        for _, _, b in self.calls:
            _chk(self.lib.vo_match_appearances_batch_dev(self.ctx.h, C.c_int(b.n_frames), C.c_void_p(b.ref_app), C.c_int(b.n_ref), None,
                                                         C.c_void_p(b.cur_app), C.c_int(b.n_cur), None, C.c_float(b.radius),
                                                         C.c_void_p(b.matches), C.c_void_p(b.counts)))

    def perturb_cur_app(self, share: float, seed: int = 0, sigma: float = 0.005):
        """Measurement helper: a share of every frame's current appearances (rows drawn at random) displaced by N(0, sigma) per
        component, in device memory -- still within the radius of their landmark, but no bitwise copy of it any more: queries
        the exact-duplicate pass of the matcher leaves open.  Applied to the arrays as they are (shares add up over calls)."""
        n, F = self.n_cur, self.F
        d = self._in[1]
        a = np.zeros((F, n, 10), np.float32)
        self.ctx.d2h(a, d)
        rng = np.random.default_rng(seed)
        k = int(share * n)
        for f in range(F):
            idx = rng.choice(n, k, replace=False)
            a[f, idx] += rng.normal(0, sigma, (k, 10)).astype(np.float32)
        self.ctx.h2d(d, a)

    def counts(self):
        raw = np.zeros(3 * self.F, np.int32)
        self.ctx.d2h(raw, self.d_counts)
        c = np.zeros((3, self.F), np.int32)
        for lo, hi, _ in self.calls:
            c[:, lo:hi] = raw[3 * lo:3 * hi].reshape(3, hi - lo)
        return c

    def poses(self):
        T = np.zeros((self.F, 16), np.float32)
        self.ctx.d2h(T, self.d_poses)
        return np.ascontiguousarray(T.reshape(self.F, 4, 4).transpose(0, 2, 1))

    def stats(self):
        s = np.zeros((self.F, 4), np.float32)
        self.ctx.d2h(s, self.d_stats)
        return s

    def fetch(self, what, f):
        c = self.counts()
        spec = {"match": (self.d_matches, c[0, f], 2, np.int32), "join": (self.d_joined, c[1, f], 2, np.int32),
                "tri_xyz": (self.d_tri_xyz, c[2, f], 3, np.float32), "tri_pairs": (self.d_tri_pairs, c[2, f], 2, np.int32),
                "tri_app": (self.d_tri_app, c[2, f], 10, np.float32)}[what]
        out = np.zeros((int(spec[1]), spec[2]), spec[3])
        if len(out):
            self.ctx.d2h(out, spec[0] + f * self.q * spec[2] * 4)
        return out

    def close(self):
        for d in self._in + [self.d_matches, self.d_joined, self.d_stats, self.d_tri_xyz,
                             self.d_tri_pairs, self.d_counts] + ([self.d_tri_app] if self.d_tri_app else []) + \
                ([self.d_moved] if self.d_moved else []) + ([self.d_X] if self.d_X else []) + \
                ([self.d_poses] if self._own_poses else []):
            self.ctx.free(d)


class _FrameSizes(C.Structure):
    """vo_frame_sizes of include/vo_hip.h"""
    _fields_ = [("n_ref", C.c_void_p), ("n_cur", C.c_void_p), ("n_model_pairs", C.c_void_p)]


def _pad_stack(arrs, cap, width, dtype):
    out = np.zeros((len(arrs), max(cap, 1), width), dtype)
    for i, a in enumerate(arrs):
        a = np.asarray(a, dtype).reshape(-1, width)
        out[i, : len(a)] = a
    return out


def match_batch_ragged(ctx: Context, apps1, apps2, radius: float = 0.1):
    """compute_correspondences_images (vo_complete.cpp:12-49) for many pairs of appearance sets of DIFFERENT sizes in one call
    (vo_match_appearances_batch_dev): returns the list of (n_i, 2) int32 pair arrays.  The up-front matching of a sequence."""
    F = len(apps1)
    assert F == len(apps2)
    if F == 0:
        return []
    n1 = np.array([len(a) for a in apps1], np.int32); n2 = np.array([len(a) for a in apps2], np.int32)
    cap1, cap2 = int(max(n1.max(), 1)), int(max(n2.max(), 1))
    q = min(cap1, cap2)
    d_a1, d_a2 = ctx.to_device(_pad_stack(apps1, cap1, 10, np.float32)), ctx.to_device(_pad_stack(apps2, cap2, 10, np.float32))
    d_n1, d_n2 = ctx.to_device(n1), ctx.to_device(n2)
    d_out, d_cnt = ctx.alloc(F * q * 8), ctx.alloc(F * 4)
    try:
        _chk(ctx.lib.vo_match_appearances_batch_dev(ctx.h, C.c_int(F), C.c_void_p(d_a1), C.c_int(cap1), C.c_void_p(d_n1),
                                                    C.c_void_p(d_a2), C.c_int(cap2), C.c_void_p(d_n2), C.c_float(radius),
                                                    C.c_void_p(d_out), C.c_void_p(d_cnt)))
        cnt = np.zeros(F, np.int32); ctx.d2h(cnt, d_cnt)
        out = np.zeros((F, q, 2), np.int32); ctx.d2h(out, d_out)
    finally:
        for d in (d_a1, d_a2, d_n1, d_n2, d_out, d_cnt):
            ctx.free(d)
    return [out[f, : cnt[f]].copy() for f in range(F)]


def frames_batch_ragged(ctx: Context, frames, K, cam, n_iters: int = 50, kernel_threshold: float = 10000.0, radius: float = 0.1,
                        keep_outliers: bool = False, X_prev=None):
    """The loop body of vo_complete.cpp:150-179 for many frames of DIFFERENT sizes in ONE call (vo_frames_batch_ragged_dev).
    frames: dicts with ref_app, cur_app, ref_pts, cur_pts, model, model_pairs; cam = (rows, cols, z_near, z_far).
    Returns per frame a dict(matches, joined, pose, stats, tri_xyz, tri_pairs)."""
    F = len(frames)
    keys = (("ref_app", np.float32, 10), ("cur_app", np.float32, 10), ("ref_pts", np.float32, 2), ("cur_pts", np.float32, 2),
            ("model", np.float32, 3), ("model_pairs", np.int32, 2))
    n = {k: np.array([len(np.asarray(f[k]).reshape(-1, w)) for f in frames], np.int32) for k, _, w in keys}
    assert np.array_equal(n["ref_app"], n["ref_pts"]) and np.array_equal(n["cur_app"], n["cur_pts"])
    cap = {k: int(max(v.max(), 1)) for k, v in n.items()}
    q = min(cap["ref_app"], cap["cur_app"])
    dev = {k: ctx.to_device(_pad_stack([f[k] for f in frames], cap[k], w, dt)) for k, dt, w in keys}
    d_n = {k: ctx.to_device(n[k]) for k in ("ref_app", "cur_app", "model_pairs")}
    a = ctx.alloc
    out = dict(matches=a(F * q * 8), joined=a(F * q * 8), moved=a(F * cap["model"] * 12), poses=a(F * 64), stats=a(F * 16),
               tri_xyz=a(F * q * 12), tri_pairs=a(F * q * 8), counts=a(3 * F * 4))
    d_X = ctx.to_device(np.ascontiguousarray(np.stack([_colmajor(X, 4) for X in X_prev]))) if X_prev is not None else None
    b = _FrameBatch()
    b.n_frames, b.n_ref, b.n_cur, b.n_model, b.n_model_pairs = F, cap["ref_app"], cap["cur_app"], cap["model"], cap["model_pairs"]
    b.ref_app, b.cur_app, b.ref_pts, b.cur_pts = dev["ref_app"], dev["cur_app"], dev["ref_pts"], dev["cur_pts"]
    b.model, b.model_pairs, b.X_prev = dev["model"], dev["model_pairs"], d_X
    b.rows, b.cols, b.z_near, b.z_far = (int(x) for x in cam)
    b.K[:] = _colmajor(K, 3).tolist()
    b.kernel_threshold, b.keep_outliers, b.n_iters, b.radius = kernel_threshold, int(keep_outliers), n_iters, radius
    b.matches, b.joined, b.model_moved, b.poses, b.stats = out["matches"], out["joined"], out["moved"], out["poses"], out["stats"]
    b.tri_xyz, b.tri_pairs, b.tri_app, b.counts = out["tri_xyz"], out["tri_pairs"], None, out["counts"]
    sz = _FrameSizes(d_n["ref_app"], d_n["cur_app"], d_n["model_pairs"])
    try:
        _chk(ctx.lib.vo_frames_batch_ragged_dev(ctx.h, C.byref(b), C.byref(sz)))
        cnt = np.zeros((3, F), np.int32); ctx.d2h(cnt, out["counts"])
        m = np.zeros((F, q, 2), np.int32); ctx.d2h(m, out["matches"])
        j = np.zeros((F, q, 2), np.int32); ctx.d2h(j, out["joined"])
        T = np.zeros((F, 16), np.float32); ctx.d2h(T, out["poses"])
        st = np.zeros((F, 4), np.float32); ctx.d2h(st, out["stats"])
        xyz = np.zeros((F, q, 3), np.float32); ctx.d2h(xyz, out["tri_xyz"])
        tp = np.zeros((F, q, 2), np.int32); ctx.d2h(tp, out["tri_pairs"])
    finally:
        for d in list(dev.values()) + list(d_n.values()) + list(out.values()) + ([d_X] if d_X else []):
            ctx.free(d)
    return [dict(matches=m[f, : cnt[0, f]].copy(), joined=j[f, : cnt[1, f]].copy(), pose=T[f].reshape(4, 4).T.copy(), stats=st[f].copy(),
                 tri_xyz=xyz[f, : cnt[2, f]].copy(), tri_pairs=tp[f, : cnt[2, f]].copy()) for f in range(F)]


class SequencePipeline:
    """A whole sequence in the manner of vo_complete.cpp:97-181 with everything resident in HBM
    (SURVEY 8(d) config 3): all measurement files are uploaded up front; the first pair is matched,
    initialised by vo_estimate_transform (the only host round trip) and triangulated; every later
    frame is match -> join -> X_curr * model -> n_iters x oneRound from the identity -> triangulate,
    chained through device-side counts and the solver's device-side pose.  keep_map: the loop body's map upkeep
    (map.update(history * triangulated_pc), history = history * pose^-1: vo_complete.cpp:145-147,175-176) runs inside
    the chain too, on the device (vo_map_*)."""

    def __init__(self, ctx: Context, seq: dict, n_iters: int = 100, kernel_threshold: float = 10000.0,
                 keep_appearance: bool = False, matches: list | None = None, overlap_match: bool = False, exact: bool = False,
                 prematch: bool = False, keep_map: bool = False, map_capacity: int | None = None):
        """prematch: the matcher depends on the appearances alone, so when the whole sequence is on hand -- as it
        is for vo_complete, which reads its measurement files from a directory -- all F-1 consecutive pairs are matched by ONE
        vo_match_appearances_batch_dev call at start() (frames of different sizes, per-frame tree choice as in the single
        call), and the chain reads frame t's pairs and count where that call left them.  Same pairs in the same order.
        exact: the solver in reference-order arithmetic (vo_picp_set_exact): the whole chain is then bit-identical to the
        float32 CPU restatement of the loop started from the same first relative pose.
        matches: optional precomputed appearance matches, matches[t-1] = (n,2) int32 pairs
        (idx in frame t-1, idx in frame t) for t = 1..F-1 -- e.g. computed up front, sharded over several
        GPUs (dist.gather_ragged); the chain then skips its own matcher launches.
        overlap_match: the matcher of frame t+1 depends on the appearances alone; run it on a second
        context (its own HIP stream), one frame ahead into a double buffer, ordered against the chain by
        events.  Results are identical; on MI355X at 50k points it is SLOWER than the single-stream chain
        (1.35 k vs 1.52 k frames/s): the matcher's waves take issue slots from the latency-bound solver
        rounds, so it is off by default and kept as a measured option."""
        self.ctx, self.lib = ctx, ctx.lib
        self.n_iters = n_iters
        fr = seq["frames"]
        self.F = len(fr)
        assert self.F >= 2, "need at least two measurement sets"
        self.n = [len(f["pts"]) for f in fr]
        self.cap = max(max(self.n), 1)
        self.off = np.concatenate([[0], np.cumsum(self.n)]).astype(np.int64)
        self.K = _colmajor(seq["K"], 3)
        self.cam = (int(seq["rows"]), int(seq["cols"]), int(seq["z_near"]), int(seq["z_far"]))
        up, a = ctx.to_device, ctx.alloc
        self.d_pts = up(np.ascontiguousarray(np.concatenate([f["pts"] for f in fr]), np.float32))
        self.d_app = up(np.ascontiguousarray(np.concatenate([f["app"] for f in fr]), np.float32))
        cap, F = self.cap, self.F
        self.d_m, self.d_j, self.d_model_t = a(cap * 8), a(cap * 8), a(cap * 12)
        self.overlap = bool(overlap_match) and matches is None and not prematch
        self.prematch = bool(prematch) and matches is None
        if self.prematch:
            # the batched call strides the frames by the capacity: a padded copy of the appearances ([F][cap][10]; the pairs
            # (t-1, t) are then two views of it, one frame apart), the sizes, and room for every frame's pairs and count
            pad = np.zeros((self.F, self.cap, 10), np.float32)
            for t, f in enumerate(fr):
                pad[t, : self.n[t]] = f["app"]
            self.d_app_pad = up(pad)
            self.d_n_all = up(np.asarray(self.n, np.int32))
            self.d_pm, self.d_pm_cnt = a((self.F - 1) * self.cap * 8), a((self.F - 1) * 4)
        if self.overlap:
            self.ctx2 = Context(ctx.device)                   # second stream of the same device
            self.d_mb = [self.d_m, a(cap * 8)]               # matches of frame t live in buffer t % 2
            self.ev_matched = [Event(ctx), Event(ctx)]       # recorded on ctx2 after match(t)
            self.ev_free = [Event(ctx), Event(ctx)]          # recorded on ctx after frame t consumed its buffer
            self._free_recorded = [False, False]
        self.pre = None
        if matches is not None:
            assert len(matches) == self.F - 1
            ms = [np.ascontiguousarray(np.asarray(m, np.int32).reshape(-1, 2)) for m in matches]
            self.pre_n = [len(m) for m in ms]
            self.pre_off = np.concatenate([[0], np.cumsum(self.pre_n)]).astype(np.int64)
            self.pre = up(np.concatenate(ms + [np.zeros((1, 2), np.int32)]))
        self.d_tri_xyz, self.d_tri_pairs = a(F * cap * 12), a(F * cap * 8)      # frame t's cloud at slice t
        self.d_tri_app = a(F * cap * 40) if (keep_appearance or keep_map) else 0
        self.map = None
        if keep_map:
            from .api import Map
            self.map = Map(ctx, int(map_capacity if map_capacity is not None else min(int(self.off[-1]), 8 * cap)))
        self.d_counts = a(3 * F * 4)                                            # [t] = (n_match, n_join, n_tri)
        c0 = np.zeros((F, 3), np.int32)
        if self.pre is not None:
            c0[1:, 0] = self.pre_n
        ctx.h2d(self.d_counts, c0)
        self.d_traj = a(F * 64)
        self.d_ident = up(np.eye(4, dtype=np.float32))
        h = C.c_void_p()
        _chk(self.lib.vo_picp_create(ctx.h, C.byref(h)))
        self.solver = h
        _chk(self.lib.vo_picp_set_camera(h, *map(C.c_int, self.cam), _ptr(self.K), _ptr(_colmajor(np.eye(4), 4))))
        _chk(self.lib.vo_picp_set_kernel_threshold(h, C.c_float(kernel_threshold)))
        _chk(self.lib.vo_picp_set_exact(h, C.c_int(1 if exact else 0)))
        p = C.c_void_p()
        _chk(self.lib.vo_picp_pose_dev_ptr(h, C.byref(p)))
        self.d_pose = p.value
        self.X0 = None

    # device addresses of frame t's inputs / outputs
    def _pts(self, t): return C.c_void_p(self.d_pts + 8 * int(self.off[t]))
    def _app(self, t): return C.c_void_p(self.d_app + 40 * int(self.off[t]))
    def _cnt(self, t, i):
        if i == 0 and self.prematch:
            return C.c_void_p(self.d_pm_cnt + 4 * (t - 1))
        return C.c_void_p(self.d_counts + 4 * (3 * t + i))
    def _xyz(self, t): return C.c_void_p(self.d_tri_xyz + 12 * self.cap * t)
    def _pairs(self, t): return C.c_void_p(self.d_tri_pairs + 8 * self.cap * t)
    def _tapp(self, t): return C.c_void_p(self.d_tri_app + 40 * self.cap * t) if self.d_tri_app else None

    def _m(self, t):
        """device address of frame t's matches"""
        if self.overlap:
            return C.c_void_p(self.d_mb[t % 2])
        if self.prematch:
            return C.c_void_p(self.d_pm + 8 * self.cap * (t - 1))
        return C.c_void_p(self.d_m if self.pre is None else self.pre + 8 * int(self.pre_off[t - 1]))

    def _match_ahead(self, t):
        """overlap mode: enqueue the matcher of frame t on the second context"""
        b = t % 2
        if self._free_recorded[b]:
            self.ev_free[b].wait(self.ctx2)          # frame t-2 must have finished reading this buffer
        _chk(self.lib.vo_match_appearances_dev(self.ctx2.h, self._app(t - 1), C.c_int(self.n[t - 1]), self._app(t),
                                               C.c_int(self.n[t]), C.c_float(0.1), self._m(t), self._cnt(t, 0)))
        self.ev_matched[b].record(self.ctx2)

    def _release(self, t):
        if self.overlap:
            self.ev_free[t % 2].record(self.ctx)
            self._free_recorded[t % 2] = True

    def match_all(self):
        """prematch mode: every consecutive pair of the sequence in one call (asynchronous)"""
        _chk(self.lib.vo_match_appearances_batch_dev(
            self.ctx.h, C.c_int(self.F - 1), C.c_void_p(self.d_app_pad), C.c_int(self.cap), C.c_void_p(self.d_n_all),
            C.c_void_p(self.d_app_pad + 40 * self.cap), C.c_int(self.cap), C.c_void_p(self.d_n_all + 4), C.c_float(0.1),
            C.c_void_p(self.d_pm), C.c_void_p(self.d_pm_cnt)))

    def _match(self, t):
        if self.pre is not None or self.prematch:
            return                                  # matched up front: pairs and count are already in place
        if self.overlap:
            self.ev_matched[t % 2].wait(self.ctx)    # enqueued earlier by _match_ahead(t)
            return
        _chk(self.lib.vo_match_appearances_dev(self.ctx.h, self._app(t - 1), C.c_int(self.n[t - 1]), self._app(t),
                                               C.c_int(self.n[t]), C.c_float(0.1), C.c_void_p(self.d_m), self._cnt(t, 0)))

    def _triangulate(self, t, X_host):
        nq = min(self.n[t - 1], self.n[t])
        _chk(self.lib.vo_triangulate_dev(self.ctx.h, _ptr(self.K), _ptr(X_host) if X_host is not None else None,
                                         None if X_host is not None else C.c_void_p(self.d_pose),
                                         self._m(t), C.c_int(nq), self._cnt(t, 0),
                                         self._pts(t - 1), C.c_int(self.n[t - 1]), self._pts(t), C.c_int(self.n[t]),
                                         self._app(t) if self.d_tri_app else None, self._xyz(t), self._pairs(t),
                                         self._tapp(t), self._cnt(t, 2)))

    def initialise(self):
        """first pair: match, epipolar initialisation (host, once per sequence), triangulate (vo_complete.cpp:121-132)"""
        self._match(1)
        X = np.zeros(16, np.float32)
        # pairs, count and both images as they lie in device memory: no copy of the pairs to the host (vo_estimate_transform_dev)
        _chk(self.lib.vo_estimate_transform_dev(self.ctx.h, _ptr(self.K), self._m(1), C.c_int(min(self.n[0], self.n[1])),
                                                self._cnt(1, 0), self._pts(0), C.c_int(self.n[0]), self._pts(1),
                                                C.c_int(self.n[1]), _ptr(X)))
        self.X0 = X
        self._triangulate(1, X)
        self._release(1)
        self.ctx.h2d(self.d_traj, np.eye(4, dtype=np.float32))
        self.ctx.h2d(self.d_traj + 64, X)
        if self.map is not None:                          # vo_complete.cpp:145-146
            self.map.clear()
            self._map_update(1, None)
            self.map.history_reset_dev(self.d_traj + 64)

    def step(self, t):
        """frame t >= 2 (vo_complete.cpp:150-179); asynchronous"""
        nq, nq_prev = min(self.n[t - 1], self.n[t]), min(self.n[t - 2], self.n[t - 1])
        if self.overlap and t + 1 < self.F:
            self._match_ahead(t + 1)                 # one frame ahead, on the second stream
        self._match(t)
        _chk(self.lib.vo_join_correspondences_dev(self.ctx.h, self._m(t), C.c_int(nq), self._cnt(t, 0),
                                                  self._pairs(t - 1), C.c_int(nq_prev), self._cnt(t - 1, 2),
                                                  C.c_int(self.n[t - 1]), C.c_void_p(self.d_j), self._cnt(t, 1)))
        _chk(self.lib.vo_transform_points_dev(self.ctx.h, _ptr(self.X0) if t == 2 else None,
                                              None if t == 2 else C.c_void_p(self.d_pose), self._xyz(t - 1),
                                              C.c_int(nq_prev), self._cnt(t - 1, 2), C.c_void_p(self.d_model_t)))
        # the capacity (not the live count) is what sizes the solver's grid: the same graph serves every frame
        _chk(self.lib.vo_picp_set_points_dev(self.solver, C.c_void_p(self.d_model_t), C.c_int(self.cap), self._pts(t),
                                             C.c_int(self.n[t])))
        _chk(self.lib.vo_picp_set_pose_dev(self.solver, C.c_void_p(self.d_ident)))
        _chk(self.lib.vo_picp_solve_dev(self.solver, C.c_void_p(self.d_j), C.c_int(self.cap), self._cnt(t, 1),
                                        C.c_int(0), C.c_int(self.n_iters)))
        _chk(self.lib.vo_picp_get_pose_dev(self.solver, C.c_void_p(self.d_traj + 64 * t)))
        self._triangulate(t, None)
        if self.map is not None:                          # vo_complete.cpp:175-176
            self._map_update(t, self.map.history_dev)
            self.map.history_step_dev(self.d_pose)
        self._release(t)

    def _map_update(self, t, d_T16):
        self.map.update_dev(self._xyz(t).value, self._tapp(t).value, min(self.n[t - 1], self.n[t]), self._cnt(t, 2).value, d_T16)

    def start(self):
        """first pair (and, in overlap mode, the matchers of frames 1 and 2 on the second stream)"""
        if self.overlap:
            self._free_recorded = [False, False]
            self.ctx.synchronize()                   # nothing of an earlier run may still read the match buffers
            self._match_ahead(1)
            if self.F > 2:
                self._match_ahead(2)
        if self.prematch:
            self.match_all()
        self.initialise()

    def run(self):
        self.start()
        for t in range(2, self.F):
            self.step(t)

    # -- results (synchronise) ----------------------------------------------------------------
    def trajectory(self):
        T = np.zeros((self.F, 16), np.float32)
        self.ctx.d2h(T, self.d_traj)
        return np.ascontiguousarray(T.reshape(self.F, 4, 4).transpose(0, 2, 1))

    def counts(self):
        c = np.zeros((self.F, 3), np.int32)
        self.ctx.d2h(c, self.d_counts)
        if self.prematch:
            m = np.zeros(self.F - 1, np.int32)
            self.ctx.d2h(m, self.d_pm_cnt)
            c[1:, 0] = m
        return c

    def cloud(self, t):
        """triangulated points of frame t (in the frame of camera t), their (idx in frame t, k) pairs and,
        when kept, appearances"""
        n = int(self.counts()[t, 2])
        xyz, pairs = np.zeros((n, 3), np.float32), np.zeros((n, 2), np.int32)
        app = np.zeros((n, 10), np.float32) if self.d_tri_app else None
        if n:
            self.ctx.d2h(xyz, self._xyz(t).value); self.ctx.d2h(pairs, self._pairs(t).value)
            if app is not None:
                self.ctx.d2h(app, self._tapp(t).value)
        return xyz, pairs, app

    def stats(self):
        ci, co, ni = C.c_float(), C.c_float(), C.c_int()
        _chk(self.lib.vo_picp_get_stats(self.solver, C.byref(ci), C.byref(co), C.byref(ni)))
        return ci.value, co.value, ni.value

    def close(self):
        if self.map is not None:
            self.map.close()
            self.map = None
        if self.solver:
            self.lib.vo_picp_destroy(self.solver)
            self.solver = None
        if self.overlap:
            self.ctx.synchronize(); self.ctx2.synchronize()
            for e in self.ev_matched + self.ev_free:
                e.close()
            self.ctx.free(self.d_mb[1])
            self.ctx2.close()
            self.overlap = False
        for d in (self.d_pts, self.d_app, self.d_m, self.d_j, self.d_model_t, self.d_tri_xyz, self.d_tri_pairs,
                  self.d_counts, self.d_traj, self.d_ident) + ((self.d_tri_app,) if self.d_tri_app else ()) + \
                ((self.pre,) if self.pre is not None else ()) + \
                ((self.d_app_pad, self.d_n_all, self.d_pm, self.d_pm_cnt) if self.prematch else ()):
            self.ctx.free(d)
