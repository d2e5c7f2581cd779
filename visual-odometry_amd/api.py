"""ctypes binding of libvo_hip.so, shaped like the reference's interface for the
projective-ICP path: `Camera` (camera.h:12-63), `PICPSolver` (picp_solver.h:18-79),
`triangulate_points` (utils.h:131-160), `compute_correspondences_images` and
`extract_correspondences_world` (vo_complete.cpp:12-66).

The production host side of this repository is C++ (include/vo/*.hpp); this
module exists so that the parity tests and bench.py can drive the very same
C ABI from Python.  There is no Python/NumPy implementation of any operator
here: if the shared library or a gfx950 device is missing, everything raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VO_HIP_LIB selects another build of the same library (e.g. the stamped diagnostic build)
LIB_PATH = os.environ.get("VO_HIP_LIB") or os.path.join(_HERE, "libvo_hip.so")

VO_OK = 0


class VoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvo_hip error {code}: {msg}")
        self.code = code


_lib = None


def load_library():
    """dlopen libvo_hip.so (built by __graft_entry__.build / make -C csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.vo_last_error.restype = C.c_char_p
        _lib.vo_ctx_stream.restype = C.c_void_p
    return _lib


def _chk(code):
    if code != VO_OK:
        raise VoError(code, load_library().vo_last_error().decode(errors="replace"))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


def _i32pairs(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(-1, 2))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _colmajor(M, n):
    return np.ascontiguousarray(np.asarray(M, dtype=np.float32).reshape(n, n).T).ravel()


class Context:
    """One (device, stream).  `stream` may be a raw hipStream_t (int), e.g.
    torch.cuda.current_stream().cuda_stream; None lets the library make one."""

    def __init__(self, device: int = 0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        _chk(self.lib.vo_ctx_create(C.c_int(device), C.c_void_p(stream or 0), C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.vo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _chk(self.lib.vo_ctx_synchronize(self.h))

    @property
    def stream(self):
        return self.lib.vo_ctx_stream(self.h)

    def device_info(self):
        name = C.create_string_buffer(128)
        ncu = C.c_int()
        _chk(self.lib.vo_ctx_device_info(self.h, name, C.c_int(128), C.byref(ncu)))
        return name.value.decode(), ncu.value

    # raw device memory (for the *_dev entry points)
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _chk(self.lib.vo_dev_alloc(self.h, C.c_size_t(nbytes), C.byref(p)))
        return p.value

    def free(self, dptr: int):
        _chk(self.lib.vo_dev_free(self.h, C.c_void_p(dptr)))

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        _chk(self.lib.vo_memcpy_h2d(self.h, C.c_void_p(dptr), _ptr(arr), C.c_size_t(arr.nbytes)))

    def d2h(self, arr: np.ndarray, dptr: int):
        assert arr.flags["C_CONTIGUOUS"]
        _chk(self.lib.vo_memcpy_d2h(self.h, _ptr(arr), C.c_void_p(dptr), C.c_size_t(arr.nbytes)))

    def to_device(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = self.alloc(max(arr.nbytes, 16))
        self.h2d(p, arr)
        return p


class Event:
    """A point in one context's stream that another context can wait for (vo_event_*)."""

    def __init__(self, ctx: Context):
        self.lib = ctx.lib
        h = C.c_void_p()
        _chk(self.lib.vo_event_create(ctx.h, C.byref(h)))
        self.h = h

    def record(self, ctx: Context):
        _chk(self.lib.vo_event_record(self.h, ctx.h))

    def wait(self, ctx: Context):
        """make later work on `ctx` wait for the recorded point (the host does not block)"""
        _chk(self.lib.vo_ctx_wait_event(ctx.h, self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.vo_event_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class Camera:
    """Pinhole camera, camera.h:12-63.  K is 3x3, pose (world in camera) 4x4."""

    def __init__(self, rows=100, cols=100, z_near=0, z_far=10, camera_matrix=None,
                 world_in_camera_pose=None, ctx: Context | None = None):
        self._rows, self._cols = int(rows), int(cols)
        self._z_near, self._z_far = int(z_near), int(z_far)     # ints, camera.h:18-19
        self._K = np.eye(3, dtype=np.float32) if camera_matrix is None else _f32(camera_matrix, (3, 3)).copy()
        self._T = np.eye(4, dtype=np.float32) if world_in_camera_pose is None else _f32(world_in_camera_pose, (4, 4)).copy()
        self._ctx = ctx

    def rows(self): return self._rows
    def cols(self): return self._cols
    def cameraMatrix(self): return self._K
    def worldInCameraPose(self): return self._T
    def setWorldInCameraPose(self, T): self._T = _f32(T, (4, 4)).copy()

    def copy(self):
        return Camera(self._rows, self._cols, self._z_near, self._z_far, self._K, self._T, self._ctx)

    def projectPoints(self, world_points, keep_indices=False):
        """camera.cpp:16-37.  Returns (image_points, num_points_inside)."""
        ctx = self._ctx or default_context()
        w = _f32(world_points, (-1, 3))
        n = len(w)
        out = np.empty((max(n, 1), 2), dtype=np.float32)
        n_out, n_in = C.c_int(), C.c_int()
        _chk(ctx.lib.vo_project_points(ctx.h, C.c_int(self._rows), C.c_int(self._cols), C.c_int(self._z_near),
                                       C.c_int(self._z_far), _ptr(_colmajor(self._K, 3)), _ptr(_colmajor(self._T, 4)),
                                       _ptr(w), C.c_int(n), C.c_int(int(keep_indices)), _ptr(out),
                                       C.byref(n_out), C.byref(n_in)))
        return out[: n_out.value].copy(), n_in.value

    def projectPoint(self, world_point):
        """camera.h:25-37 through the batched kernel.  Returns (ok, uv)."""
        uv, n_in = self.projectPoints(np.asarray(world_point, dtype=np.float32).reshape(1, 3), keep_indices=True)
        return n_in == 1, uv[0]


class PICPSolver:
    """picp_solver.h:18-79 on the GPU.  oneRound() enqueues and returns; camera(),
    numInliers(), chiInliers(), chiOutliers() synchronise."""

    def __init__(self, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        h = C.c_void_p()
        _chk(self.lib.vo_picp_create(self.ctx.h, C.byref(h)))
        self.h = h
        self._cam = Camera(ctx=self.ctx)

    def close(self):
        if getattr(self, "h", None):
            self.lib.vo_picp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init(self, camera: Camera, world_points, image_points):
        """picp_solver.cpp:16-23"""
        self._cam = camera.copy()
        _chk(self.lib.vo_picp_set_camera(self.h, C.c_int(camera.rows()), C.c_int(camera.cols()),
                                         C.c_int(camera._z_near), C.c_int(camera._z_far),
                                         _ptr(_colmajor(camera.cameraMatrix(), 3)),
                                         _ptr(_colmajor(camera.worldInCameraPose(), 4))))
        w = _f32(world_points, (-1, 3))
        z = _f32(image_points, (-1, 2))
        _chk(self.lib.vo_picp_set_points(self.h, _ptr(w), C.c_int(len(w)), _ptr(z), C.c_int(len(z))))

    def kernelThreshold(self):
        t = C.c_float()
        _chk(self.lib.vo_picp_get_kernel_threshold(self.h, C.byref(t)))
        return t.value

    def setKernelThreshold(self, thr):
        _chk(self.lib.vo_picp_set_kernel_threshold(self.h, C.c_float(thr)))

    def _pairs(self, correspondences):
        # converted on every call (no copy for a C-contiguous int32 array): the library compares the whole array
        # with its GPU copy, so an in-place edit between two rounds is honoured (picp_solver.cpp:62)
        return _i32pairs(correspondences)

    def setExact(self, on=True):
        """reference-order arithmetic (vo_picp_set_exact): bit-identical to the reference's float32 loop"""
        _chk(self.lib.vo_picp_set_exact(self.h, C.c_int(int(bool(on)))))

    def graphInfo(self):
        """(use_graph, cached graphs, failed captures) of this handle: vo_picp_graph_info"""
        u, n, f = C.c_int(), C.c_int(), C.c_int()
        _chk(self.lib.vo_picp_graph_info(self.h, C.byref(u), C.byref(n), C.byref(f)))
        return u.value, n.value, f.value

    def chainInfo(self):
        """(rounds whose finishing launch is still to come, oneRound calls enqueued ahead of their comparison, how many
        of those were repeated): vo_picp_chain_info"""
        o, a, b = C.c_int(), C.c_ulonglong(), C.c_ulonglong()
        _chk(self.lib.vo_picp_chain_info(self.h, C.byref(o), C.byref(a), C.byref(b)))
        return o.value, a.value, b.value

    def setCorrespondences(self, correspondences):
        p = _i32pairs(correspondences)
        _chk(self.lib.vo_picp_set_correspondences(self.h, _ptr(p), C.c_int(len(p))))

    def rounds(self, keep_outliers=False, n_iters=1):
        _chk(self.lib.vo_picp_rounds(self.h, C.c_int(int(keep_outliers)), C.c_int(n_iters)))

    def oneRound(self, correspondences, keep_outliers=False) -> bool:
        """picp_solver.cpp:98-112; pairs are (measurement index, world index)."""
        p = self._pairs(correspondences)
        _chk(self.lib.vo_picp_one_round(self.h, _ptr(p), C.c_int(len(p)), C.c_int(int(keep_outliers))))
        return True   # min_num_inliers is 0 in the reference: oneRound cannot return false

    def solve(self, correspondences, keep_outliers=False, n_iters=1):
        p = self._pairs(correspondences)
        _chk(self.lib.vo_picp_solve(self.h, _ptr(p), C.c_int(len(p)), C.c_int(int(keep_outliers)), C.c_int(n_iters)))

    def camera(self) -> Camera:
        T = np.zeros(16, dtype=np.float32)
        _chk(self.lib.vo_picp_get_pose(self.h, _ptr(T)))
        cam = self._cam.copy()
        cam.setWorldInCameraPose(T.reshape(4, 4).T)
        return cam

    def _stats(self):
        ci, co, ni = C.c_float(), C.c_float(), C.c_int()
        _chk(self.lib.vo_picp_get_stats(self.h, C.byref(ci), C.byref(co), C.byref(ni)))
        return ci.value, co.value, ni.value

    def chiInliers(self): return self._stats()[0]
    def chiOutliers(self): return self._stats()[1]
    def numInliers(self): return self._stats()[2]

    def system(self):
        """(H with damping, b) of the last round, as the reference leaves _H/_b."""
        H = np.zeros(36, dtype=np.float32)
        b = np.zeros(6, dtype=np.float32)
        _chk(self.lib.vo_picp_get_system(self.h, _ptr(H), _ptr(b)))
        return H.reshape(6, 6).T.copy(), b


class Map:
    """PointCloudVector<3> `map` of vo_complete on the GPU (vo_map_*): update() is PointCloud.h:52-66, the history
    isometry vo_complete.cpp:145-147,175-176."""

    def __init__(self, ctx: Context | None = None, capacity: int = 0):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        h = C.c_void_p()
        _chk(self.lib.vo_map_create(self.ctx.h, C.c_int(capacity), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.vo_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        _chk(self.lib.vo_map_clear(self.h))

    def update(self, points, appearances, T=None):
        """map.update(T * cloud) from host arrays"""
        p = _f32(points, (-1, 3))
        a = _f32(appearances, (-1, 10))
        assert len(p) == len(a)
        _chk(self.lib.vo_map_update(self.h, _ptr(p), _ptr(a), C.c_int(len(p)), _ptr(_colmajor(T, 4)) if T is not None else None))

    def update_dev(self, d_xyz, d_app, n_max, d_n=None, d_T16=None):
        _chk(self.lib.vo_map_update_dev(self.h, C.c_void_p(d_xyz), C.c_void_p(d_app), C.c_int(n_max),
                                        C.c_void_p(d_n) if d_n else None, C.c_void_p(d_T16) if d_T16 else None))

    def history_reset_dev(self, d_X16):
        _chk(self.lib.vo_map_history_reset_dev(self.h, C.c_void_p(d_X16)))

    def history_step_dev(self, d_X16):
        _chk(self.lib.vo_map_history_step_dev(self.h, C.c_void_p(d_X16)))

    @property
    def history_dev(self):
        p = C.c_void_p()
        _chk(self.lib.vo_map_history_dev_ptr(self.h, C.byref(p)))
        return p.value

    def history(self):
        T = np.zeros(16, np.float32)
        _chk(self.lib.vo_map_get_history(self.h, _ptr(T)))
        return T.reshape(4, 4).T.copy()

    def transform(self, T):
        _chk(self.lib.vo_map_transform(self.h, _ptr(_colmajor(T, 4))))

    def __len__(self):
        n = C.c_int()
        _chk(self.lib.vo_map_size(self.h, C.byref(n)))
        return n.value

    def read(self):
        """(points (n, 3), appearances (n, 10)) in entry order"""
        n = len(self)
        p = np.zeros((max(n, 1), 3), np.float32)
        a = np.zeros((max(n, 1), 10), np.float32)
        m = C.c_int()
        _chk(self.lib.vo_map_read(self.h, _ptr(p), _ptr(a), C.c_int(n), C.byref(m)))
        return p[:n].copy(), a[:n].copy()


def triangulate_points(k, X, correspondences, p1_img, p2_img, app2=None, want_pairs=True,
                       ctx: Context | None = None):
    """utils.cpp:51-134.  Returns (triangulated, correspondences_new, appearances)."""
    ctx = ctx or default_context()
    pairs = _i32pairs(correspondences)
    a = _f32(p1_img, (-1, 2))
    b = _f32(p2_img, (-1, 2))
    n = len(pairs)
    xyz = np.zeros((max(n, 1), 3), dtype=np.float32)
    outp = np.zeros((max(n, 1), 2), dtype=np.int32)
    app = _f32(app2, (-1, 10)) if app2 is not None else None
    oapp = np.zeros((max(n, 1), 10), dtype=np.float32) if app is not None else None
    n_out = C.c_int()
    nul = C.c_void_p(0)
    _chk(ctx.lib.vo_triangulate(ctx.h, _ptr(_colmajor(k, 3)), _ptr(_colmajor(X, 4)), _ptr(pairs), C.c_int(n),
                                _ptr(a), C.c_int(len(a)), _ptr(b), C.c_int(len(b)),
                                _ptr(app) if app is not None else nul, _ptr(xyz),
                                _ptr(outp) if want_pairs else nul, _ptr(oapp) if oapp is not None else nul,
                                C.byref(n_out)))
    m = n_out.value
    return xyz[:m].copy(), (outp[:m].copy() if want_pairs else None), (oapp[:m].copy() if oapp is not None else None)


def compute_correspondences_images(appearances1, appearances2, radius=0.1, ctx: Context | None = None):
    """vo_complete.cpp:12-49 -> pairs (ref_idx, curr_idx)."""
    ctx = ctx or default_context()
    a1 = _f32(appearances1, (-1, 10))
    a2 = _f32(appearances2, (-1, 10))
    out = np.zeros((max(min(len(a1), len(a2)), 1), 2), dtype=np.int32)
    n_out = C.c_int()
    _chk(ctx.lib.vo_match_appearances(ctx.h, _ptr(a1), C.c_int(len(a1)), _ptr(a2), C.c_int(len(a2)),
                                      C.c_float(radius), _ptr(out), C.byref(n_out)))
    return out[: n_out.value].copy()


def extract_correspondences_world(correspondences_imgs, correspondences_world, ctx: Context | None = None):
    """vo_complete.cpp:52-66 -> pairs (curr_idx, world_idx)."""
    ctx = ctx or default_context()
    a = _i32pairs(correspondences_imgs)
    b = _i32pairs(correspondences_world)
    out = np.zeros((max(len(a), 1), 2), dtype=np.int32)
    n_out = C.c_int()
    _chk(ctx.lib.vo_join_correspondences(ctx.h, _ptr(a), C.c_int(len(a)), _ptr(b), C.c_int(len(b)), _ptr(out),
                                         C.byref(n_out)))
    return out[: n_out.value].copy()


def transform_points(X, points, ctx: Context | None = None):
    """Isometry3f * point set, PointCloud.h:77-82."""
    ctx = ctx or default_context()
    p = _f32(points, (-1, 3))
    out = np.zeros_like(p)
    _chk(ctx.lib.vo_transform_points(ctx.h, _ptr(_colmajor(X, 4)), _ptr(p), C.c_int(len(p)), _ptr(out)))
    return out


def estimate_transform(k, correspondences, p1_img, p2_img, ctx: Context | None = None):
    """epipolar_utils.cpp:176-213: relative pose (first camera in the frame of the second) from >= 8
    image correspondences; the cheirality vote runs the GPU triangulation kernel."""
    ctx = ctx or default_context()
    pairs = _i32pairs(correspondences)
    a = _f32(p1_img, (-1, 2))
    b = _f32(p2_img, (-1, 2))
    X = np.zeros(16, dtype=np.float32)
    _chk(ctx.lib.vo_estimate_transform(ctx.h, _ptr(_colmajor(k, 3)), _ptr(pairs), C.c_int(len(pairs)), _ptr(a),
                                       C.c_int(len(a)), _ptr(b), C.c_int(len(b)), _ptr(X)))
    return X.reshape(4, 4).T.copy()


def radius_search(tree_appearances, query_appearances, radius=0.1, ctx: Context | None = None):
    """TreeNode_::fullSearch (eigen_kdtree.h:56-71) for every query: list of int32 arrays, one per query,
    with the indices of ALL tree points closer than `radius` (ascending; the library's order is unspecified)."""
    ctx = ctx or default_context()
    t = _f32(tree_appearances, (-1, 10))
    q = _f32(query_appearances, (-1, 10))
    offsets = np.zeros(len(q) + 1, dtype=np.int32)
    cap = max(2 * len(q), 16)
    while True:
        idx = np.zeros(cap, dtype=np.int32)
        n_total = C.c_int()
        rc = ctx.lib.vo_radius_search(ctx.h, _ptr(t), C.c_int(len(t)), _ptr(q), C.c_int(len(q)), C.c_float(radius),
                                      _ptr(offsets), _ptr(idx), C.c_int(cap), C.byref(n_total))
        if rc == -1 and n_total.value > cap:            # not enough room: the call reports what it needs
            cap = n_total.value
            continue
        _chk(rc)
        break
    return [np.sort(idx[offsets[i]:offsets[i + 1]]) for i in range(len(q))]


class KdTree:
    """The reference's PCA kd-tree in its approximate modes (eigen_kdtree.h:18-52,75-85; vo_kdtree_*): built on the
    host like the TreeNode_ constructor, queried on the GPU.  bestMatchFull / fullSearch are tree-independent:
    compute_correspondences_images / radius_search."""

    def __init__(self, points_appearances, max_points_in_leaf=20, ctx: Context | None = None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        p = _f32(points_appearances, (-1, 10))
        h = C.c_void_p()
        _chk(self.lib.vo_kdtree_create(self.ctx.h, _ptr(p), C.c_int(len(p)), C.c_int(max_points_in_leaf), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.vo_kdtree_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        n, nodes, leaves = C.c_int(), C.c_int(), C.c_int()
        _chk(self.lib.vo_kdtree_info(self.h, C.byref(n), C.byref(nodes), C.byref(leaves)))
        return n.value, nodes.value, leaves.value

    def bestMatchFast(self, queries, norm=0.1):
        """index of the best point of each query's leaf within `norm`, or -1 (eigen_kdtree.h:75-85)"""
        q = _f32(queries, (-1, 10))
        out = np.full(max(len(q), 1), -1, dtype=np.int32)
        _chk(self.lib.vo_kdtree_best_match_fast(self.h, _ptr(q), C.c_int(len(q)), C.c_float(norm), _ptr(out)))
        return out[:len(q)].copy()

    def fastSearch(self, queries, norm=0.1):
        """per query the points of its leaf within `norm`, in leaf order (eigen_kdtree.h:40-52)"""
        q = _f32(queries, (-1, 10))
        offsets = np.zeros(len(q) + 1, dtype=np.int32)
        cap = max(2 * len(q), 16)
        while True:
            idx = np.zeros(cap, dtype=np.int32)
            n_total = C.c_int()
            rc = self.lib.vo_kdtree_fast_search(self.h, _ptr(q), C.c_int(len(q)), C.c_float(norm), _ptr(offsets), _ptr(idx),
                                                C.c_int(cap), C.byref(n_total))
            if rc == -1 and n_total.value > cap:
                cap = n_total.value
                continue
            _chk(rc)
            break
        return [idx[offsets[i]:offsets[i + 1]].copy() for i in range(len(q))]
