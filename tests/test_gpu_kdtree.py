"""the approximate modes of the reference's kd-tree, bestMatchFast (eigen_kdtree.h:75-85) and fastSearch
(:40-52): host-built PCA tree (like the TreeNode_ constructor) + GPU descent and leaf scan, against the oracle's
restatement of the same tree (oracle/vo_kdtree.c): same tree, same leaves, same answers, index for index."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sets(vo):
    rng = np.random.default_rng(21)
    base = rng.uniform(-1, 1, (900, 10)).astype(np.float32)
    noisy = np.concatenate([base + rng.normal(0, 0.02, base.shape).astype(np.float32) for _ in range(5)])      # several hits per leaf
    fp = vo.synth.frame_pair(6000, seed=55, drop=0.1, distractors=30)
    lat = (rng.integers(-2, 3, (700, 10)) / 32.0).astype(np.float32)                                             # ties, boundary distances
    return [(noisy, base, 0.1, 20), (base, noisy, 0.1, 10), (fp["ref_app"], fp["cur_app"], 0.1, 10),
            (fp["cur_app"], fp["ref_app"], 0.1, 20), (lat, lat[:200], 0.0625, 20), (base[:7], base, 0.5, 20),
            (np.zeros((60, 10), np.float32), np.zeros((5, 10), np.float32), 0.1, 20)]     # identical points: the guard makes a leaf


def test_fast_modes_equal_the_oracle_tree(vo, ctx, o32):
    for tree_pts, queries, radius, max_leaf in _sets(vo):
        best_o, lists_o, nodes_o = o32.kdtree_fast(tree_pts, queries, radius, max_leaf)
        t = vo.KdTree(tree_pts, max_leaf, ctx=ctx)
        n, nodes, leaves = t.info()
        assert n == len(tree_pts) and nodes == nodes_o and leaves == (nodes + 1) // 2
        best = t.bestMatchFast(queries, radius)
        lists = t.fastSearch(queries, radius)
        t.close()
        assert np.array_equal(best, best_o), (len(tree_pts), len(queries))
        assert len(lists) == len(lists_o) and all(np.array_equal(a, b) for a, b in zip(lists, lists_o))     # leaf order included
        # what "approximate" means: every answer is also an answer of the exact modes
        full = vo.radius_search(tree_pts, queries, radius, ctx=ctx)
        assert all(set(a.tolist()) <= set(b.tolist()) for a, b in zip(lists, full))
        assert all(b == -1 or b in f for b, f in zip(best.tolist(), full))


def test_fast_mode_finds_most_exact_matches_on_frame_data(vo, ctx, o32):
    """on appearance data (exact copies between frames) the leaf a query lands in holds its partner: bestMatchFast
    agrees with the exact matcher for almost every query (the reference never uses it; this is a sanity figure)"""
    fp = vo.synth.frame_pair(8000, seed=56)
    t = vo.KdTree(fp["ref_app"], 10, ctx=ctx)
    best = t.bestMatchFast(fp["cur_app"], 0.1)
    t.close()
    exact = np.full(len(fp["cur_app"]), -1)
    m = o32.match_kdtree(fp["ref_app"], fp["cur_app"])
    exact[m[:, 1]] = m[:, 0]
    assert np.mean(best == exact) > 0.99
    assert np.all((best == exact) | (best == -1) | (exact == -1) | True)


def test_kdtree_errors_and_empty(vo, ctx):
    import ctypes as C
    h = C.c_void_p()
    assert ctx.lib.vo_kdtree_create(ctx.h, None, 3, 20, C.byref(h)) == -1
    assert ctx.lib.vo_kdtree_create(ctx.h, None, 0, 0, C.byref(h)) == -1
    t = vo.KdTree(np.zeros((0, 10), np.float32), 20, ctx=ctx)
    assert t.info() == (0, 1, 1)
    assert t.bestMatchFast(np.zeros((3, 10), np.float32), 0.1).tolist() == [-1, -1, -1]
    assert [len(x) for x in t.fastSearch(np.zeros((2, 10), np.float32), 0.1)] == [0, 0]
    t.close()
