"""the approximate kd-tree modes depend on the split directions, i.e. on the eigen-solver
(eigen_covariance.h:35-43: Eigen's SelfAdjointEigenSolver).  Product (csrc/kdtree.hip + the host build in capi.hip) and oracle
(oracle/vo_kdtree.c) share ONE stand-in for it, a cyclic Jacobi in double -- so their agreement alone (tests/test_gpu_kdtree.py)
is a sibling comparison.  Here the oracle's tree is held to an INDEPENDENT restatement: the same construction written in numpy with
LAPACK's `eigh` (another algorithm: tridiagonalisation + implicit QL/QR) for the eigenvectors -- mean and covariance accumulated
sequentially in float32 as computeMeanAndCovariance does (:5-30), the largest eigenvector with the library's sign convention, the
two-pointer partition of split.h:8-34, recursion while a node holds >= max_points_in_leaf points.  Every query must land in the
same leaf with the same leaf order: bestMatchFast (eigen_kdtree.h:75-85) and fastSearch (:40-52) equal index for index."""
import numpy as np

F = np.float32


def _direction(v):
    """mean (float32, sequential) and largest eigenvector of the float32 covariance, via LAPACK"""
    k = len(v)
    m = np.cumsum(v, axis=0, dtype=F)[-1]                                     # sequential float32 sums, in array order
    outer = (v[:, :, None] * v[:, None, :]).astype(F)
    cov = np.cumsum(outer, axis=0, dtype=F)[-1]
    ik = F(1.0 / k)
    m = (m * ik).astype(F)
    cov = (cov * ik).astype(F)
    cov = (cov - np.outer(m, m).astype(F)).astype(F)
    cov = (cov * (F(k) / F(k - 1))).astype(F)
    w, vec = np.linalg.eigh(cov.astype(np.float64))
    n = vec[:, int(np.argmax(w))]
    big = int(np.argmax(np.abs(n)))                                            # sign: the component of largest magnitude positive
    if n[big] < 0:
        n = -n
    return m, n.astype(F)


def _plane_dist(p, mean, normal):
    s = F(0)
    for i in range(10):                                                        # left to right, float32, unfused
        s = F(s + F(F(p[i] - mean[i]) * normal[i]))
    return s


def _build(pts, idx, begin, end, max_leaf, nodes):
    node = dict(begin=begin, end=end, left=None, right=None)
    nodes.append(node)
    if end - begin < max_leaf:
        return node
    mean, normal = _direction(pts[begin:end])
    node["mean"], node["normal"] = mean, normal
    lower, upper = begin, end                                                  # split.h:8-34
    while lower != upper:
        if _plane_dist(pts[lower], mean, normal) < 0:
            lower += 1
        else:
            pts[[lower, upper - 1]] = pts[[upper - 1, lower]]
            idx[[lower, upper - 1]] = idx[[upper - 1, lower]]
            upper -= 1
    if upper == begin or upper == end:
        return node
    node["left"] = _build(pts, idx, begin, upper, max_leaf, nodes)
    node["right"] = _build(pts, idx, upper, end, max_leaf, nodes)
    return node


def _query(root, pts, idx, q, radius):
    n = root
    while n["left"] is not None or n["right"] is not None:
        n = n["left"] if _plane_dist(q, n["mean"], n["normal"]) < 0 else n["right"]
    best, best_d, hits = -1, F(radius) * F(radius), []
    r2 = F(radius) * F(radius)
    for j in range(n["begin"], n["end"]):
        d = F(0)
        for k in range(10):
            t = F(pts[j, k] - q[k]); d = F(d + F(t * t))
        if d < r2:
            hits.append(int(idx[j]))
        if d < best_d:
            best_d, best = d, int(idx[j])
    return best, hits


def test_oracle_tree_equals_an_eigh_based_restatement(o32):
    rng = np.random.default_rng(21)
    for n, nq, max_leaf, radius in ((900, 300, 20, 0.3), (2500, 400, 10, 0.25)):
        base = rng.uniform(-1, 1, (n, 10)).astype(F)
        base[:, 3] *= F(2.5)                                                   # one dominant direction at the root, then the rest
        q = (base[rng.permutation(n)[:nq]] + rng.normal(0, 0.03, (nq, 10))).astype(F)
        best_o, lists_o, n_nodes_o = o32.kdtree_fast(base, q, radius, max_leaf)
        pts, idx, nodes = base.copy(), np.arange(n), []
        root = _build(pts, idx, 0, n, max_leaf, nodes)
        assert len(nodes) == n_nodes_o
        n_hit = 0
        for i in range(nq):
            b, hits = _query(root, pts, idx, q[i], radius)
            assert b == best_o[i], (n, i)
            assert hits == lists_o[i].tolist(), (n, i)                          # same leaf, same order inside it
            n_hit += len(hits)
        assert n_hit > nq // 2
