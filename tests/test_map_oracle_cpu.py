"""The oracle's map (oracle/vo_pipeline.py) held to the reference's loops as written -- no GPU.
`Map` is a dictionary of first occurrences; `literal_update` is PointCloudVector::update (PointCloud.h:52-66) line by line:
two nested loops, the first entry whose appearance compares equal (operator== on ten floats) gets the point, otherwise the
pair is appended.  The shortcut must leave the same entries in the same order on everything == distinguishes from bytes:
duplicates inside a cloud, -0 against +0, NaN rows.  The float32 isometry helpers against double."""
import numpy as np

from oracle import vo_pipeline as P


def _cloud(rng, n, pool):
    idx = rng.integers(0, len(pool), n)
    a = pool[idx].copy()
    z = rng.random(a.shape) < 0.15
    a[z] = np.where(rng.random(int(z.sum())) < 0.5, np.float32(0.0), np.float32(-0.0))
    bad = rng.random(n) < 0.03
    a[bad, rng.integers(0, 10, int(bad.sum()))] = np.nan
    return rng.normal(0, 3, (n, 3)).astype(np.float32), a.astype(np.float32)


def test_first_occurrence_dictionary_equals_the_double_loop():
    rng = np.random.default_rng(1)
    for trial in range(6):
        pool = np.round(rng.uniform(-1, 1, (60, 10)), 1).astype(np.float32)
        pool[:8] = np.where(rng.random((8, 10)) < 0.5, np.float32(0.0), np.float32(-0.0))     # rows of zeros of either sign: one class
        m = P.Map()
        lp, la = [], []
        for n in (0, 1, 40, 150, 7):
            pts, app = _cloud(rng, n, pool)
            m.update(list(pts), list(app))
            P.literal_update(lp, la, list(pts), list(app))
            assert len(m.pts) == len(lp)
            assert np.array(m.pts, np.float32).tobytes() == np.array(lp, np.float32).tobytes()
            assert np.array(m.app, np.float32).tobytes() == np.array(la, np.float32).tobytes()      # bits: the first occurrence's row stays
        nan_rows = int(np.isnan(np.array(la, np.float32).reshape(-1, 10)).any(axis=1).sum())
        assert nan_rows >= 1 and len(lp) < 198 + nan_rows                                           # classes were found again, NaN rows never


def test_zero_signs_and_nan_follow_operator_equal():
    a = np.zeros(10, np.float32); b = a.copy(); b[3] = np.float32(-0.0)
    c = a.copy(); c[5] = np.nan
    m = P.Map()
    m.update([np.float32([1, 1, 1])], [a])
    m.update([np.float32([2, 2, 2])], [b])            # -0 == +0: the same entry, its point replaced, its bits kept
    assert len(m.pts) == 1 and m.pts[0][0] == 2 and np.array(m.app[0]).tobytes() == a.tobytes()
    m.update([np.float32([3, 3, 3]), np.float32([4, 4, 4])], [c, c])      # NaN != NaN: appended, both
    assert len(m.pts) == 3


def test_float32_isometry_helpers():
    rng = np.random.default_rng(2)
    for _ in range(50):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        X = np.eye(4); X[:3, :3] = q * np.sign(np.linalg.det(q)); X[:3, 3] = rng.uniform(-2, 2, 3)
        Y = np.eye(4); q2, _ = np.linalg.qr(rng.normal(size=(3, 3))); Y[:3, :3] = q2 * np.sign(np.linalg.det(q2)); Y[:3, 3] = rng.uniform(-2, 2, 3)
        X32, Y32 = X.astype(np.float32), Y.astype(np.float32)
        inv = P.iso_inv32(X32)
        assert inv.dtype == np.float32 and np.abs(inv.astype(np.float64) - np.linalg.inv(X32.astype(np.float64))).max() < 2e-6
        assert np.array_equal(inv[:3, :3], X32[:3, :3].T)                       # the rotation is a transpose, exactly
        prod = P.iso_mul32(X32, Y32)
        assert prod.dtype == np.float32 and np.abs(prod.astype(np.float64) - X32.astype(np.float64) @ Y32.astype(np.float64)).max() < 2e-6
        assert np.array_equal(prod[3], np.float32([0, 0, 0, 1]))
