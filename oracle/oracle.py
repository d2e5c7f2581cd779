"""ctypes front-end of the CPU oracle (oracle/libvo_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package.

PARITY PINNED BY KNOWN ANSWERS, UNPINNED AT THE BIT LEVEL (see vo_oracle_impl.h): the
reference has no golden vectors and needs Eigen3, which the image lacks, so there is
no reference build to compare bits with; its data directory carries the ground truth,
and the oracle reproduces it through the reference's own known-association scenarios
and the README metrics (oracle/vo_pipeline.py, tests/test_known_answers_cpu.py).

All matrices cross this interface as numpy arrays in the usual mathematical
(row, col) indexing; they are flattened column-major (Eigen's default layout)
before they reach C.  `bits` selects ref32 (float, the reference's arithmetic)
or ref64 (double, the arbiter).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libvo_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("vo_oracle.c", "vo_oracle_impl.h", "vo_kdtree.c")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def native_lib():
    """the same sources compiled -march=native ON THIS HOST (bench.py's second, separately labelled CPU baseline)"""
    subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
    return C.CDLL(os.path.join(_HERE, "libvo_oracle_native.so"))


@dataclass
class Camera:
    """Mirror of the reference Camera's state (camera.h:55-61)."""
    rows: int = 100
    cols: int = 100
    z_near: int = 0
    z_far: int = 10
    K: np.ndarray = field(default_factory=lambda: np.eye(3))
    T: np.ndarray = field(default_factory=lambda: np.eye(4))


def _cam_struct(real):
    class S(C.Structure):
        _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("z_near", C.c_int), ("z_far", C.c_int),
                    ("K", real * 9), ("T", real * 16)]
    return S


def _picp_struct(real, cam_t):
    class S(C.Structure):
        _fields_ = [("cam", cam_t), ("kernel_threshold", real), ("damping", real),
                    ("min_num_inliers", C.c_int), ("world", C.c_void_p), ("meas", C.c_void_p),
                    ("H", real * 36), ("b", real * 6), ("chi_inliers", real),
                    ("chi_outliers", real), ("num_inliers", C.c_int)]
    return S


class Oracle:
    def __init__(self, bits: int = 32, library=None):
        assert bits in (32, 64)
        self.bits = bits
        self.dt = np.float32 if bits == 32 else np.float64
        self.real = C.c_float if bits == 32 else C.c_double
        self.pfx = "vo32_" if bits == 32 else "vo64_"
        self.cam_t = _cam_struct(self.real)
        self.picp_t = _picp_struct(self.real, self.cam_t)
        self.L = library if library is not None else lib()

    # -- helpers ---------------------------------------------------------
    def _f(self, name):
        return getattr(self.L, self.pfx + name)

    def _arr(self, a, shape=None):
        a = np.ascontiguousarray(np.asarray(a, dtype=self.dt))
        if shape is not None:
            a = a.reshape(shape)
        return a

    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def _cam(self, cam: Camera):
        s = self.cam_t()
        s.rows, s.cols, s.z_near, s.z_far = int(cam.rows), int(cam.cols), int(cam.z_near), int(cam.z_far)
        K = np.asarray(cam.K, dtype=self.dt).reshape(3, 3)
        T = np.asarray(cam.T, dtype=self.dt).reshape(4, 4)
        s.K[:] = K.ravel(order="F").tolist()
        s.T[:] = T.ravel(order="F").tolist()
        return s

    @staticmethod
    def _pairs(p):
        return np.ascontiguousarray(np.asarray(p, dtype=np.int32).reshape(-1, 2))

    # -- Camera ----------------------------------------------------------
    def project_points(self, cam: Camera, world, keep_indices: bool = False):
        w = self._arr(world, (-1, 3))
        out = np.empty((len(w), 2), dtype=self.dt)
        n_out = C.c_int(0)
        s = self._cam(cam)
        f = self._f("project_points")
        f.restype = C.c_int
        n_in = f(C.byref(s), self._p(w), C.c_int(len(w)), C.c_int(int(keep_indices)), self._p(out),
                 C.byref(n_out))
        return out[: n_out.value].copy(), int(n_in)

    # -- PICPSolver ------------------------------------------------------
    def picp_solve(self, cam: Camera, world, meas, corr, n_iters: int, kernel_threshold=1000.0,
                   keep_outliers: bool = False, trace: bool = True):
        """n_iters x oneRound from cam.T; returns the final pose and, when
        trace, per-iteration H (pre-damping), b, (chi_in, chi_out, n_in), T."""
        w = self._arr(world, (-1, 3))
        z = self._arr(meas, (-1, 2))
        cp = self._pairs(corr)
        s = self.picp_t()
        self._f("picp_ctor")(C.byref(s))
        cs = self._cam(cam)
        self._f("picp_init")(C.byref(s), C.byref(cs), self._p(w), self._p(z))
        s.kernel_threshold = kernel_threshold
        tH = np.zeros((n_iters, 36), dtype=self.dt) if trace else None
        tb = np.zeros((n_iters, 6), dtype=self.dt) if trace else None
        ts = np.zeros((n_iters, 3), dtype=self.dt) if trace else None
        tT = np.zeros((n_iters, 16), dtype=self.dt) if trace else None
        nul = C.c_void_p(0)
        self._f("picp_solve")(C.byref(s), self._p(cp), C.c_int(len(cp)), C.c_int(int(keep_outliers)),
                              C.c_int(n_iters),
                              self._p(tH) if trace else nul, self._p(tb) if trace else nul,
                              self._p(ts) if trace else nul, self._p(tT) if trace else nul)
        res = {
            "T": np.array(s.cam.T[:], dtype=self.dt).reshape(4, 4, order="F"),
            "chi_inliers": float(s.chi_inliers), "chi_outliers": float(s.chi_outliers),
            "num_inliers": int(s.num_inliers),
        }
        if trace:
            res["H"] = tH.reshape(n_iters, 6, 6).transpose(0, 2, 1).copy()  # col-major -> (r,c)
            res["b"] = tb
            res["stats"] = ts
            res["T_trace"] = tT.reshape(n_iters, 4, 4).transpose(0, 2, 1).copy()
        return res

    def picp_solve_raw(self, cam: Camera, world, meas, corr, n_iters: int, kernel_threshold=1000.0,
                       keep_outliers: bool = False):
        """n_iters x oneRound from cam.T; per round what the solver holds afterwards, untouched: H WITH the
        damping (as _H is left, picp_solver.cpp:102), b, (chi_in, chi_out, n_in), pose -- for bitwise checks."""
        w = self._arr(world, (-1, 3))
        z = self._arr(meas, (-1, 2))
        cp = self._pairs(corr)
        s = self.picp_t()
        self._f("picp_ctor")(C.byref(s))
        cs = self._cam(cam)
        self._f("picp_init")(C.byref(s), C.byref(cs), self._p(w), self._p(z))
        s.kernel_threshold = kernel_threshold
        tH = np.zeros((n_iters, 36), dtype=self.dt)
        tb = np.zeros((n_iters, 6), dtype=self.dt)
        ts = np.zeros((n_iters, 3), dtype=self.dt)
        tT = np.zeros((n_iters, 16), dtype=self.dt)
        self._f("picp_solve_raw")(C.byref(s), self._p(cp), C.c_int(len(cp)), C.c_int(int(keep_outliers)),
                                  C.c_int(n_iters), self._p(tH), self._p(tb), self._p(ts), self._p(tT))
        return {"H": tH.reshape(n_iters, 6, 6).transpose(0, 2, 1).copy(), "b": tb, "stats": ts,
                "T": tT.reshape(n_iters, 4, 4).transpose(0, 2, 1).copy()}

    def picp_solve_mt(self, cam: Camera, world, meas, corr, n_iters: int, n_threads: int,
                      kernel_threshold=1000.0, keep_outliers: bool = False):
        """all-cores baseline (float32 only): per-thread partial sums, see vo_oracle.c"""
        assert self.bits == 32
        w = self._arr(world, (-1, 3))
        z = self._arr(meas, (-1, 2))
        cp = self._pairs(corr)
        s = self.picp_t()
        self._f("picp_ctor")(C.byref(s))
        cs = self._cam(cam)
        self._f("picp_init")(C.byref(s), C.byref(cs), self._p(w), self._p(z))
        s.kernel_threshold = kernel_threshold
        f = self.L.vo32_picp_solve_mt
        f.restype = C.c_int
        used = f(C.byref(s), self._p(cp), C.c_int(len(cp)), C.c_int(int(keep_outliers)), C.c_int(n_iters),
                 C.c_int(n_threads))
        return {"T": np.array(s.cam.T[:], dtype=self.dt).reshape(4, 4, order="F"), "threads": int(used),
                "chi_inliers": float(s.chi_inliers), "chi_outliers": float(s.chi_outliers),
                "num_inliers": int(s.num_inliers),
                "H": np.array(s.H[:], dtype=self.dt).reshape(6, 6, order="F"), "b": np.array(s.b[:], dtype=self.dt)}

    def ldlt_solve(self, A, rhs):
        A = np.asarray(A, dtype=self.dt)
        n = A.shape[0]
        Af = np.ascontiguousarray(A.ravel(order="F"))
        r = self._arr(rhs)
        x = np.zeros(n, dtype=self.dt)
        self._f("ldlt_solve")(C.c_int(n), self._p(Af), self._p(r), self._p(x))
        return x

    def v2t_euler(self, v):
        v = self._arr(v)
        T = np.zeros(16, dtype=self.dt)
        self._f("v2t_euler")(self._p(v), self._p(T))
        return T.reshape(4, 4, order="F")

    # -- triangulation ---------------------------------------------------
    def triangulate(self, K, X, corr, p1, p2, app2=None, want_pairs=True):
        Kf = np.ascontiguousarray(np.asarray(K, dtype=self.dt).reshape(3, 3).ravel(order="F"))
        Xf = np.ascontiguousarray(np.asarray(X, dtype=self.dt).reshape(4, 4).ravel(order="F"))
        cp = self._pairs(corr)
        a = self._arr(p1, (-1, 2))
        b = self._arr(p2, (-1, 2))
        n = len(cp)
        xyz = np.zeros((n, 3), dtype=self.dt)
        pairs = np.zeros((n, 2), dtype=np.int32)
        app = self._arr(app2, (-1, 10)) if app2 is not None else None
        oapp = np.zeros((n, 10), dtype=self.dt) if app is not None else None
        f = self._f("triangulate_points")
        f.restype = C.c_int
        nul = C.c_void_p(0)
        m = f(self._p(Kf), self._p(Xf), self._p(cp), C.c_int(n), self._p(a), self._p(b),
              self._p(app) if app is not None else nul, self._p(xyz),
              self._p(pairs) if want_pairs else nul, self._p(oapp) if oapp is not None else nul)
        return xyz[:m].copy(), pairs[:m].copy(), (oapp[:m].copy() if oapp is not None else None)

    # -- matcher / join / transform ---------------------------------------
    def match(self, a1, a2, radius=0.1):
        x = self._arr(a1, (-1, 10))
        y = self._arr(a2, (-1, 10))
        out = np.zeros((min(len(x), len(y)), 2), dtype=np.int32)
        f = self._f("match")
        f.restype = C.c_int
        n = f(self._p(x), C.c_int(len(x)), self._p(y), C.c_int(len(y)), self.real(radius), self._p(out))
        return out[:n].copy()

    def match_kdtree(self, a1, a2, radius=0.1, max_leaf=10, timing=False):
        """The reference's own matcher: PCA kd-tree + bestMatchFull (float32 only)."""
        x = np.ascontiguousarray(a1, dtype=np.float32).reshape(-1, 10)
        y = np.ascontiguousarray(a2, dtype=np.float32).reshape(-1, 10)
        out = np.zeros((max(min(len(x), len(y)), 1), 2), dtype=np.int32)
        tb, tq = C.c_double(), C.c_double()
        f = self.L.vo32_match_kdtree
        f.restype = C.c_int
        n = f(self._p(x), C.c_int(len(x)), self._p(y), C.c_int(len(y)), C.c_float(radius), C.c_int(max_leaf),
              self._p(out), C.byref(tb), C.byref(tq))
        res = out[:n].copy()
        return (res, tb.value, tq.value) if timing else res

    def radius_search(self, tree_app, query_app, radius=0.1, max_leaf=10, brute=False):
        """TreeNode_::fullSearch for every query (float32): list of ascending index arrays.  brute=True: the
        plain double loop instead of the kd-tree traversal."""
        t = np.ascontiguousarray(tree_app, dtype=np.float32).reshape(-1, 10)
        q = np.ascontiguousarray(query_app, dtype=np.float32).reshape(-1, 10)
        off = np.zeros(len(q) + 1, dtype=np.int32)
        f = self.L.vo32_radius_search
        f.restype = C.c_int
        cap = max(4 * len(q), 16)
        while True:
            idx = np.zeros(cap, dtype=np.int32)
            total = f(self._p(t), C.c_int(len(t)), self._p(q), C.c_int(len(q)), C.c_float(radius), C.c_int(max_leaf),
                      C.c_int(int(brute)), self._p(off), self._p(idx), C.c_int(cap))
            if total <= cap:
                break
            cap = total
        return [np.sort(idx[off[i]:off[i + 1]]) for i in range(len(q))]

    def kdtree_fast(self, tree_app, query_app, radius=0.1, max_leaf=20):
        """The approximate modes of the reference's tree (float32): (bestMatchFast index per query or -1,
        fastSearch lists per query IN LEAF ORDER, number of tree nodes)."""
        t = np.ascontiguousarray(tree_app, dtype=np.float32).reshape(-1, 10)
        q = np.ascontiguousarray(query_app, dtype=np.float32).reshape(-1, 10)
        best = np.zeros(max(len(q), 1), dtype=np.int32)
        off = np.zeros(len(q) + 1, dtype=np.int32)
        n_nodes = C.c_int()
        f = self.L.vo32_kdtree_fast
        f.restype = C.c_int
        cap = max(4 * len(q), 16)
        while True:
            idx = np.zeros(cap, dtype=np.int32)
            total = f(self._p(t), C.c_int(len(t)), self._p(q), C.c_int(len(q)), C.c_float(radius), C.c_int(max_leaf),
                      self._p(best), self._p(off), self._p(idx), C.c_int(cap), C.byref(n_nodes))
            if total <= cap:
                break
            cap = total
        return best[:len(q)].copy(), [idx[off[i]:off[i + 1]].copy() for i in range(len(q))], n_nodes.value

    def join(self, img_pairs, world_pairs, linear=False):
        a = self._pairs(img_pairs)
        b = self._pairs(world_pairs)
        out = np.zeros((len(a), 2), dtype=np.int32)
        f = self.L.vo_join_linear if linear else self.L.vo_join
        f.restype = C.c_int
        n = f(self._p(a), C.c_int(len(a)), self._p(b), C.c_int(len(b)), self._p(out))
        return out[:n].copy()

    def transform_points(self, T, pts):
        Tf = np.ascontiguousarray(np.asarray(T, dtype=self.dt).reshape(4, 4).ravel(order="F"))
        p = self._arr(pts, (-1, 3))
        out = np.zeros_like(p)
        self._f("transform_points")(self._p(Tf), self._p(p), C.c_int(len(p)), self._p(out))
        return out
