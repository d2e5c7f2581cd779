"""Oracle-side restatement of the reference's whole `vo_complete` + `evaluation`
run (vo_complete.cpp:68-187, evaluate.cpp:7-90, epipolar_utils.cpp:48-213,
files_utils.cpp), built from the C oracle's operators plus numpy for the
once-per-sequence linear algebra (SVDs).  TEST INFRASTRUCTURE ONLY.

This is what lets the oracle be checked against the only numbers the reference
publishes for this path: the README metrics on example_data (README.md:74-79).
"""
from __future__ import annotations

import os
import re

import numpy as np

from .oracle import Camera, Oracle


# ---- files_utils.cpp ----------------------------------------------------------
def read_meas(path):
    """(pts (n,2), app (n,10), ids (n,)) from meas-XXXXX.dat"""
    pts, app, ids = [], [], []
    with open(path) as f:
        lines = f.read().splitlines()[3:]
    for line in lines:
        w = line.split()
        if not w:
            continue
        ids.append(int(w[2]))
        v = [float(x) for x in w[3:15]]
        pts.append(v[:2]); app.append(v[2:])
    return (np.array(pts, np.float32).reshape(-1, 2), np.array(app, np.float32).reshape(-1, 10),
            np.array(ids, np.int64))


def read_world(path):
    a = np.loadtxt(path, dtype=np.float64)
    return a[:, 1:4].astype(np.float32), a[:, 4:14].astype(np.float32)


def read_camera(path):
    lines = open(path).read().splitlines()
    K = np.zeros((3, 3), np.float32); H = np.eye(4, dtype=np.float32); ints = {}
    i = 0
    while i < len(lines):
        w = lines[i].split()
        if not w:
            i += 1; continue
        if w[0] == "camera":
            for r in range(3):
                K[r] = [float(x) for x in lines[i + 1 + r].split()]
            i += 4; continue
        if w[0] == "cam_transform:":
            for r in range(4):
                H[r] = [float(x) for x in lines[i + 1 + r].split()]
            i += 5; continue
        if w[0] in ("z_near:", "z_far:", "width:", "height:"):
            ints[w[0][:-1]] = int(w[1])
        i += 1
    return K, H, ints


def read_gt(path):
    """evaluation_utils.cpp:3-31: (x, y, theta) -> planar isometries"""
    out = []
    for line in open(path):
        w = line.split()
        if not w:
            continue
        x, y, th = float(w[4]), float(w[5]), float(w[6])
        T = np.eye(4)
        T[:2, :2] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
        T[:2, 3] = [x, y]
        out.append(T)
    return out


# ---- epipolar_utils.cpp ---------------------------------------------------------
def normalize(p):
    mx, my = max(0.0, float(p[:, 0].max())), max(0.0, float(p[:, 1].max()))
    out = np.stack([p[:, 0] / (mx / 2) - 1, p[:, 1] / (my / 2) - 1], axis=1)
    T = np.array([[1 / (mx / 2), 0, -1], [0, 1 / (my / 2), -1], [0, 0, 1]])
    return out, T


def estimate_fundamental(corr, p1, p2):
    assert len(corr) >= 8
    a, T1 = normalize(p1.astype(np.float64)); b, T2 = normalize(p2.astype(np.float64))
    d1 = np.concatenate([a[corr[:, 0]], np.ones((len(corr), 1))], axis=1)
    d2 = np.concatenate([b[corr[:, 1]], np.ones((len(corr), 1))], axis=1)
    A = np.einsum("ni,nj->nij", d1, d2).reshape(len(corr), 9)
    Fa = np.linalg.svd(A)[2][8].reshape(3, 3)
    U, s, Vt = np.linalg.svd(Fa)
    F = U @ np.diag([s[0], s[1], 0]) @ Vt
    return T1.T @ F @ T2


def essential_to_pair(E):
    W = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], float)
    U, _, Vt = np.linalg.svd(E)
    R1 = Vt.T @ W @ U.T
    if np.linalg.det(R1) < 0:
        U, _, Vt = np.linalg.svd(-E)
        R1 = Vt.T @ W @ U.T
    R2 = Vt.T @ W.T @ U.T

    def make(R):
        ts = R @ E
        X = np.eye(4); X[:3, :3] = R; X[:3, 3] = [ts[2, 1], ts[0, 2], ts[1, 0]]
        return X
    return make(R1), make(R2)


def estimate_transform(o: Oracle, K, corr, p1, p2):
    F = estimate_fundamental(corr, p1, p2)
    E = K.astype(np.float64).T @ F @ K.astype(np.float64)
    X1, X2 = essential_to_pair(E)
    best, n_best = np.eye(4, dtype=np.float32), 0
    for X in (X1, X2):
        for sgn in (1.0, -1.0):
            Xt = X.copy(); Xt[:3, 3] *= sgn
            Xt = Xt.astype(np.float32)
            n = len(o.triangulate(K, Xt, corr, p1, p2, want_pairs=False)[0])
            if n > n_best:
                best, n_best = Xt, n
    return best


# ---- vo_complete.cpp ------------------------------------------------------------
def iso_inv(T):
    R = T[:3, :3]; t = T[:3, 3]
    X = np.eye(4, dtype=T.dtype); X[:3, :3] = R.T; X[:3, 3] = -(R.T @ t)
    return X


def _dot3_32(a0, b0, a1, b1, a2, b2):
    """a 3-term inner product in float32 as Eigen 3.4 sums it (fixed size, not vectorised): x0 + (x1 + x2)"""
    f = np.float32
    return f(f(f(a0) * f(b0)) + f(f(f(a1) * f(b1)) + f(f(a2) * f(b2))))


def iso_inv32(T):
    """Isometry3f::inverse() in float32: R^T, -(R^T t) (vo_complete.cpp:146,176)"""
    T = np.asarray(T, np.float32)
    X = np.eye(4, dtype=np.float32)
    X[:3, :3] = T[:3, :3].T
    for r in range(3):
        X[r, 3] = -_dot3_32(X[r, 0], T[0, 3], X[r, 1], T[1, 3], X[r, 2], T[2, 3])
    return X


def iso_mul32(A, B):
    """Isometry3f * Isometry3f in float32: R = Ra Rb, t = Ra tb + ta (vo_complete.cpp:176)"""
    A = np.asarray(A, np.float32); B = np.asarray(B, np.float32)
    C = np.eye(4, dtype=np.float32)
    for c in range(3):
        for r in range(3):
            C[r, c] = _dot3_32(A[r, 0], B[0, c], A[r, 1], B[1, c], A[r, 2], B[2, c])
    for r in range(3):
        C[r, 3] = np.float32(_dot3_32(A[r, 0], B[0, 3], A[r, 1], B[1, 3], A[r, 2], B[2, 3]) + np.float32(A[r, 3]))
    return C


class Map:
    """PointCloudVector::update (PointCloud.h:52-66): for every point of the cloud, in order, the first entry whose
    appearance compares equal -- operator== on ten floats: -0 equals +0, a row with a NaN equals nothing -- gets the
    point, otherwise the pair is appended.  The dictionary holds the first entry of every class, keyed by the bytes of the
    row with -0 turned into +0 (x + 0.0); NaN rows are appended without a key.  literal_update() below is the double
    loop itself, for the tests that pin this shortcut."""

    def __init__(self):
        self.pts, self.app, self.idx = [], [], {}

    def update(self, pts, app):
        if len(pts) == 0:
            return
        A = np.ascontiguousarray(np.asarray(app, np.float32).reshape(-1, 10))
        nan = np.isnan(A).any(axis=1)
        canon = np.ascontiguousarray(A + np.float32(0.0))        # -0 -> +0: rows equal under == share their bytes
        keys = canon.view("V40").ravel()                          # one 40-byte record per row
        for i, p in enumerate(pts):
            if nan[i]:
                self.pts.append(p); self.app.append(A[i])
                continue
            k = keys[i].tobytes()
            j = self.idx.get(k)
            if j is not None:
                self.pts[j] = p
            else:
                self.idx[k] = len(self.pts)
                self.pts.append(p); self.app.append(A[i])


def literal_update(map_pts, map_app, pts, app):
    """the reference's loops as written (PointCloud.h:52-66), on Python lists of float32 arrays; O(N M)"""
    for p, a in zip(pts, app):
        found = False
        for j in range(len(map_app)):
            if bool(np.all(map_app[j] == a)):          # Eigen's operator==: every component compares equal
                map_pts[j] = p
                found = True
                break
        if not found:
            map_pts.append(p); map_app.append(a)


def id_correspondences(ids_ref, ids_cur):
    """extract_correspondences_images of the known-association programs (vo_daKnown.cpp:20-35,
    initialization_real_data.cpp:20-35): pairs (ref_idx, cur_idx) with equal landmark id, in ref order; both id lists are
    ascending (the scan of the current frame stops at the first larger id), first hit wins."""
    out = []
    for i, a in enumerate(ids_ref):
        for j, b in enumerate(ids_cur):
            if b > a:
                break
            if b == a:
                out.append((i, j))
                break
    return np.array(out, np.int32).reshape(-1, 2)


def run_sequence(frames, K, rows, cols, zn, zf, rounds=100, o: Oracle | None = None, X0=None, kdtree=False,
                 keep_map=True, by_id=False):
    """The frame loop of vo_complete.cpp:97-181 on measurement sets held in memory: frames = list of (pts (n,2), app
    (n,10)).  X0: first relative pose to start the chain from (default: the epipolar initialisation).  kdtree: match
    with the reference's own PCA kd-tree restatement (oracle/vo_kdtree.c) instead of the double loop -- same pairs,
    usable at 50k points.  by_id: frames = list of (pts, ids) and the association is the known one (the loop of
    vo_daKnown.cpp:96-160, which is the same loop with ids in place of appearances).  Returns trajectory, per-frame
    (matches, joined, inliers) and, with keep_map, the map."""
    o = o or Oracle(32)
    match = (lambda a, b: o.match_kdtree(a, b)) if kdtree else (lambda a, b: o.match(a, b))
    if by_id:
        match = id_correspondences
        keep_map = False
    ref_pts, ref_app = frames[0]
    cur_pts, cur_app = frames[1]
    corr = match(ref_app, cur_app)
    X = estimate_transform(o, K, corr, ref_pts, cur_pts) if X0 is None else np.asarray(X0, np.float32).reshape(4, 4).copy()
    app_of = (lambda a: None) if by_id else (lambda a: a)         # vo_daKnown triangulates without appearances
    tri, corr_world, tri_app = o.triangulate(K, X, corr, ref_pts, cur_pts, app_of(cur_app))
    traj = [np.eye(4, dtype=np.float32), X.copy()]
    m = Map()
    if keep_map:
        m.update(tri, tri_app)
    history = iso_inv32(X)                                       # Isometry3f arithmetic, like the reference (vo_complete.cpp:146)
    X_curr = X
    ref_pts, ref_app = cur_pts, cur_app
    stats, tri_counts = [], [len(tri)]
    for cur_pts, cur_app in frames[2:]:
        corr = match(ref_app, cur_app)
        corr_world = o.join(corr, corr_world, linear=kdtree)      # (the O(C) form gives the same pairs; asserted in test_oracle)
        moved = o.transform_points(X_curr, tri)
        r = o.picp_solve(Camera(rows, cols, zn, zf, K, np.eye(4)), moved, cur_pts, corr_world, rounds, 10000.0, False,
                         trace=False)
        X_curr = r["T"]
        traj.append(X_curr.copy())
        stats.append((len(corr), len(corr_world), r["num_inliers"]))
        tri, corr_world, tri_app = o.triangulate(K, X_curr, corr, ref_pts, cur_pts, app_of(cur_app))
        tri_counts.append(len(tri))
        if keep_map:
            m.update(o.transform_points(history, tri) if len(tri) else tri, tri_app)
        history = iso_mul32(history, iso_inv32(X_curr))          # vo_complete.cpp:176
        ref_pts, ref_app = cur_pts, cur_app
    return dict(trajectory=traj, stats=stats, tri_counts=tri_counts, map=m)


def run_vo_complete(data_dir, rounds=100, o: Oracle | None = None, X0=None):
    """X0: first relative pose to start the chain from (default: the epipolar initialisation below)."""
    o = o or Oracle(32)
    files = sorted(f for f in os.listdir(data_dir) if re.search(r"^meas-\d.*\.dat$", f))
    K, H, ints = read_camera(os.path.join(data_dir, "camera.dat"))
    rows, cols, zn, zf = ints["height"], ints["width"], ints["z_near"], ints["z_far"]
    frames = [read_meas(os.path.join(data_dir, f))[:2] for f in files]
    res = run_sequence(frames, K, rows, cols, zn, zf, rounds, o, X0)
    m = res["map"]
    map_pts = o.transform_points(H, np.array(m.pts, np.float32).reshape(-1, 3))
    return dict(trajectory=res["trajectory"], map=map_pts, map_app=np.array(m.app, np.float32).reshape(-1, 10), H=H,
                stats=res["stats"])


# ---- the reference's known-association programs (src/tests/) ------------------------------------
def _dataset(data_dir):
    files = sorted(f for f in os.listdir(data_dir) if re.search(r"^meas-\d.*\.dat$", f))
    K, H, ints = read_camera(os.path.join(data_dir, "camera.dat"))
    return files, K, H, (ints["height"], ints["width"], ints["z_near"], ints["z_far"])


def run_picp_known_real(data_dir, rounds=1000, o: Oracle | None = None):
    """picp_real_data_allKnown.cpp:64-89: landmark positions (world.dat) and association (the ids of the measurement files)
    known.  Per frame: the world points are moved IN PLACE by the last estimate (first by H^-1), the solver starts from
    the identity and runs `rounds` rounds on the pairs (measurement i, landmark id_i).  Returns the relative poses, H and
    per-frame (correspondences, inliers)."""
    o = o or Oracle(32)
    files, K, H, (rows, cols, zn, zf) = _dataset(data_dir)
    world, _ = read_world(os.path.join(data_dir, "world.dat"))
    X = iso_inv(H.astype(np.float64)).astype(np.float32)
    pts = world.copy()
    traj, stats = [], []
    for f in files:
        meas, _, ids = read_meas(os.path.join(data_dir, f))
        pts = o.transform_points(X, pts)
        corr = np.stack([np.arange(len(ids)), ids], axis=1).astype(np.int32)
        r = o.picp_solve(Camera(rows, cols, zn, zf, K, np.eye(4)), pts, meas, corr, rounds, 10000.0, False, trace=False)
        X = r["T"]
        traj.append(X.copy())
        stats.append((len(corr), r["num_inliers"]))
    return dict(trajectory=traj, H=H, stats=stats)


def run_real_init(data_dir, o: Oracle | None = None):
    """initialization_real_data.cpp:78-99: relative pose of the first two frames from the known association, the
    triangulated points mapped by H.  Returns X, the pairs, the points and the landmark id of every point."""
    o = o or Oracle(32)
    files, K, H, _ = _dataset(data_dir)
    p0, _, id0 = read_meas(os.path.join(data_dir, files[0]))
    p1, _, id1 = read_meas(os.path.join(data_dir, files[1]))
    corr = id_correspondences(id0, id1)
    X = estimate_transform(o, K, corr, p0, p1)
    xyz, pairs, _ = o.triangulate(K, X, corr, p0, p1)
    return dict(X=X, corr=corr, points=o.transform_points(H, xyz), ids=id1[pairs[:, 0]], H=H, K=K, p0=p0, p1=p1)


def run_vo_da_known(data_dir, rounds=1000, o: Oracle | None = None, X0=None):
    """vo_daKnown.cpp:54-167: the whole loop with the association taken from the landmark ids."""
    o = o or Oracle(32)
    files, K, H, (rows, cols, zn, zf) = _dataset(data_dir)
    frames = []
    for f in files:
        pts, _, ids = read_meas(os.path.join(data_dir, f))
        frames.append((pts, ids))
    res = run_sequence(frames, K, rows, cols, zn, zf, rounds, o, X0, by_id=True)
    return dict(trajectory=res["trajectory"], H=H, stats=res["stats"], tri_counts=res["tri_counts"])


def gt_errors(data_dir, trajectory, H, up_to_scale=False):
    """robot trajectory of the estimate against trajectory.dat: per-pose max abs difference (4x4), after the evaluation's
    median-ratio scale correction when the estimate is only defined up to scale (evaluate.cpp:40-60)."""
    gt = read_gt(os.path.join(data_dir, "trajectory.dat"))
    est = robot_trajectory(trajectory, H)
    scale = 1.0
    if up_to_scale:
        ratio = []
        for i in range(1, len(gt)):
            Xr = np.linalg.inv(est[i - 1]) @ est[i]; Xg = np.linalg.inv(gt[i - 1]) @ gt[i]
            if np.linalg.norm(Xg[:3, 3]) > 0:
                ratio.append(np.linalg.norm(Xr[:3, 3]) / np.linalg.norm(Xg[:3, 3]))
        scale = 1.0 / sorted(ratio)[len(ratio) // 2]
    out = []
    for g, e in zip(gt, est):
        e = e.copy(); e[:3, 3] *= scale
        out.append(float(np.abs(g - e).max()))
    return np.array(out), scale


def robot_trajectory(traj, H):
    """save_trajectory (files_utils.cpp:136-153): H <- H C X_i^-1 C^-1"""
    C = H.astype(np.float64); Ci = np.linalg.inv(C)
    W = np.eye(4); out = []
    for X in traj:
        W = W @ C @ iso_inv(X.astype(np.float64)) @ Ci
        out.append(W.copy())
    return out


def evaluate(data_dir, res):
    """evaluate.cpp:18-86 -> dict of the README metrics"""
    gt = read_gt(os.path.join(data_dir, "trajectory.dat"))
    est = robot_trajectory(res["trajectory"], res["H"])
    e_th, ratio = [], []
    for i in range(1, len(gt)):
        Xr = np.linalg.inv(est[i - 1]) @ est[i]; Xg = np.linalg.inv(gt[i - 1]) @ gt[i]
        e_th.append(np.trace(np.eye(3) - Xr[:3, :3].T @ Xg[:3, :3]))
        with np.errstate(divide="ignore", invalid="ignore"):           # a zero ground-truth step gives inf, as in the reference
            ratio.append(np.linalg.norm(Xr[:3, 3]) / np.linalg.norm(Xg[:3, 3]))
    r = sorted(ratio)[len(ratio) // 2]
    scale = 1.0 / r
    rmse_pos = np.sqrt(np.mean([np.linalg.norm(g[:3, 3] - e[:3, 3] * scale) ** 2 for g, e in zip(gt, est)]))
    world, world_app = read_world(os.path.join(data_dir, "world.dat"))
    first = {}
    for j, a in enumerate(world_app):
        first.setdefault(a.tobytes(), j)
    errs = [np.linalg.norm(p.astype(np.float64) * scale - world[first[a.tobytes()]]) ** 2
            for p, a in zip(res["map"], res["map_app"]) if a.tobytes() in first]
    return dict(mean_orientation_error=float(np.mean(e_th)), median_ratio_inv=float(scale),
                rmse_position=float(rmse_pos), rmse_map=float(np.sqrt(np.mean(errs))), matched=len(errs))
