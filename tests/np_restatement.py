"""Independent float64 numpy restatement of the PICP / triangulation maths,
written from the formulas (not from the C oracle) so that the two can check
each other.  Test infrastructure only."""
import numpy as np


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def v2t_euler(v):
    cx, sx = np.cos(v[3]), np.sin(v[3])
    cy, sy = np.cos(v[4]), np.sin(v[4])
    cz, sz = np.cos(v[5]), np.sin(v[5])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rx @ Ry @ Rz
    T[:3, 3] = v[:3]
    return T


def linearize(K, T, world, meas, corr, thr, keep_outliers, rows, cols, z_near, z_far):
    K = np.asarray(K, np.float64); T = np.asarray(T, np.float64)
    pw = np.asarray(world, np.float64)[corr[:, 1]]
    z = np.asarray(meas, np.float64)[corr[:, 0]]
    pc = pw @ T[:3, :3].T + T[:3, 3]
    ok = ~((pc[:, 2] > z_far) | (pc[:, 2] < z_near))
    ph = pc @ K.T
    with np.errstate(all="ignore"):
        uv = ph[:, :2] / ph[:, 2:3]
    ok &= ~((uv[:, 0] < 0) | (uv[:, 0] > cols - 1) | (uv[:, 1] < 0) | (uv[:, 1] > rows - 1))
    e = uv - z
    chi = (e * e).sum(1)
    outl = ok & (chi > thr)
    inl = ok & ~(chi > thr)
    n = len(pc)
    Jr = np.zeros((n, 3, 6))
    Jr[:, 0, 0] = Jr[:, 1, 1] = Jr[:, 2, 2] = 1
    v = -pc
    Jr[:, 0, 4] = -v[:, 2]; Jr[:, 0, 5] = v[:, 1]
    Jr[:, 1, 3] = v[:, 2]; Jr[:, 1, 5] = -v[:, 0]
    Jr[:, 2, 3] = -v[:, 1]; Jr[:, 2, 4] = v[:, 0]
    with np.errstate(all="ignore"):
        iz = 1.0 / ph[:, 2]
    Jp = np.zeros((n, 2, 3))
    Jp[:, 0, 0] = iz; Jp[:, 1, 1] = iz
    Jp[:, 0, 2] = -ph[:, 0] * iz * iz
    Jp[:, 1, 2] = -ph[:, 1] * iz * iz
    J = Jp @ K @ Jr
    with np.errstate(all="ignore"):
        lam = np.where(outl, np.sqrt(thr / chi), 1.0)
    use = inl | (outl & bool(keep_outliers))
    w = np.where(use, lam, 0.0)
    J = np.where(use[:, None, None], J, 0.0)
    e = np.where(use[:, None], e, 0.0)
    H = np.einsum("nij,nik,n->jk", J, J, w)
    b = np.einsum("nij,ni,n->j", J, e, w)
    return H, b, float(chi[inl].sum()), float(chi[outl].sum()), int(inl.sum())


def solve(K, T0, world, meas, corr, n_iters, thr, keep_outliers, rows, cols, z_near, z_far, damping=1.0):
    T = np.asarray(T0, np.float64).copy()
    hist = []
    for _ in range(n_iters):
        H, b, ci, co, ni = linearize(K, T, world, meas, corr, thr, keep_outliers, rows, cols, z_near, z_far)
        dx = np.linalg.solve(H + damping * np.eye(6), -b)
        T = v2t_euler(dx) @ T
        hist.append((H, b, ci, co, ni, T.copy()))
    return T, hist


def triangulate(K, X, corr, p1, p2):
    K = np.asarray(K, np.float64); X = np.asarray(X, np.float64)
    # Isometry3f::inverse() is (R^T, -R^T t): not the general inverse when R is
    # only orthonormal to float32 precision
    iR = X[:3, :3].T
    t = -iR @ X[:3, 3]
    iK = np.linalg.inv(K)
    iRiK = iR @ iK
    pts, pairs = [], []
    for i1, i2 in corr:
        d1 = iK @ np.array([p1[i1, 0], p1[i1, 1], 1.0])
        d2 = iRiK @ np.array([p2[i2, 0], p2[i2, 1], 1.0])
        D = np.stack([-d1, d2], axis=1)
        ss = -np.linalg.solve(D.T @ D, D.T @ t)
        if ss[0] < 0 or ss[1] < 0:
            continue
        pairs.append((i2, len(pts)))
        pts.append(0.5 * (ss[0] * d1 + t + ss[1] * d2))
    return np.array(pts).reshape(-1, 3), np.array(pairs, dtype=np.int32).reshape(-1, 2)


def match(a1, a2, radius=0.1):
    a1 = np.asarray(a1, np.float32); a2 = np.asarray(a2, np.float32)
    tree_is_1 = len(a1) >= len(a2)
    tree, qry = (a1, a2) if tree_is_1 else (a2, a1)
    out = []
    r2 = np.float32(radius) * np.float32(radius)
    for q in range(len(qry)):
        d = ((tree.astype(np.float64) - qry[q].astype(np.float64)) ** 2).sum(1)
        j = int(np.argmin(d)) if len(d) else -1
        if j >= 0 and d[j] < r2:
            out.append((j, q) if tree_is_1 else (q, j))
    return np.array(out, dtype=np.int32).reshape(-1, 2)
