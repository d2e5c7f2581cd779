"""The map on the GPU (vo_map_*, map.hip) against the reference's own loops: PointCloudVector<3>::update
(PointCloud.h:52-66) restated literally (oracle/vo_pipeline.py: literal_update, O(N M)) and through the oracle's
first-occurrence dictionary (Map), entry for entry -- order, appearance bits, points."""
import numpy as np
import pytest

from oracle import vo_pipeline as P

pytestmark = pytest.mark.gpu


def _same(gpu_map, pts, app):
    p, a = gpu_map.read()
    assert len(p) == len(pts), (len(p), len(pts))
    want_p = np.array(pts, np.float32).reshape(-1, 3)
    want_a = np.array(app, np.float32).reshape(-1, 10)
    assert p.tobytes() == want_p.tobytes()
    assert a.tobytes() == want_a.tobytes()         # bits: the FIRST occurrence's row stays (a later -0 does not replace a +0)


def _cloud(rng, n, pool, dup=0.3, zeros=0.1, nans=0.02):
    """n rows drawn from a pool of appearances (so that clouds overlap each other and themselves), some components zeroed
    with either sign, some rows given a NaN"""
    if n == 0:
        return np.zeros((0, 3), np.float32), np.zeros((0, 10), np.float32)
    idx = rng.integers(0, len(pool), n)
    k = int(dup * n)
    idx[rng.choice(n, k, replace=False)] = idx[rng.choice(n, k)]          # duplicates inside the cloud
    a = pool[idx].copy()
    z = rng.random(a.shape) < zeros
    a[z] = np.where(rng.random(int(z.sum())) < 0.5, np.float32(0.0), np.float32(-0.0))
    bad = rng.random(n) < nans
    a[bad, rng.integers(0, 10, int(bad.sum()))] = np.nan
    return rng.normal(0, 3, (n, 3)).astype(np.float32), a.astype(np.float32)


def test_update_equals_the_literal_double_loop(vo, ctx, o32):
    rng = np.random.default_rng(11)
    pool = np.round(rng.uniform(-1, 1, (400, 10)), 1).astype(np.float32)   # coarse values: many rows share components, some whole rows
    pool[:40, :] = np.where(rng.random((40, 10)) < 0.5, np.float32(0.0), np.float32(-0.0))   # rows of zeros of either sign: ONE class
    m = vo.Map(ctx)
    lit_p, lit_a = [], []
    dic = P.Map()
    for step in range(6):
        n = [0, 1, 257, 600, 1500, 90][step]
        pts, app = _cloud(rng, n, pool)
        T = None
        if step % 2:
            T = np.eye(4, dtype=np.float32); T[:3, :3] = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32); T[:3, 3] = [0.5, -2, 1]
        m.update(pts, app, T)
        moved = o32.transform_points(T, pts) if (T is not None and n) else pts
        P.literal_update(lit_p, lit_a, list(moved), list(app))
        dic.update(list(moved), list(app))
        _same(m, lit_p, lit_a)
        _same(m, dic.pts, dic.app)
    assert len(m) < sum([0, 1, 257, 600, 1500, 90])            # (classes were found again)
    assert np.isnan(np.array(lit_a)).any(axis=1).sum() > 5     # NaN rows were appended, every one
    m.clear()
    assert len(m) == 0
    m.close()


def test_map_grows_and_keeps_its_entries(vo, ctx):
    rng = np.random.default_rng(12)
    m = vo.Map(ctx, capacity=1024)
    dic = P.Map()
    for step in range(7):
        n = 3000
        app = rng.uniform(-1, 1, (n, 10)).astype(np.float32)
        if step:
            app[: n // 2] = prev[rng.choice(len(prev), n // 2, replace=False)]      # half of them seen before
        pts = rng.normal(0, 1, (n, 3)).astype(np.float32)
        m.update(pts, app)
        dic.update(list(pts), list(app))
        prev = np.array(dic.app, np.float32)
    assert len(dic.pts) > 8 * 1024
    _same(m, dic.pts, dic.app)
    m.close()


def test_map_at_50k_rows_per_update_with_device_arrays(vo, ctx, o32):
    """the size of a synthetic frame: 50 000 rows per update from device memory, a live-row count in device memory, the
    history isometry kept on the device (reset from a pose, stepped by another, applied to the cloud)"""
    rng = np.random.default_rng(13)
    n = 50000
    m = vo.Map(ctx, capacity=4 * n)
    dic = P.Map()
    X1 = vo.synth.random_isometry(rng, 0.3, 0.5).astype(np.float32)
    X2 = vo.synth.random_isometry(rng, 0.3, 0.5).astype(np.float32)
    d_X1, d_X2 = ctx.to_device(np.ascontiguousarray(X1.T)), ctx.to_device(np.ascontiguousarray(X2.T))
    m.history_reset_dev(d_X1)
    hist = P.iso_inv32(X1)
    assert np.array_equal(m.history(), hist)
    world = rng.uniform(-1, 1, (3 * n, 10)).astype(np.float32)
    for step in range(4):
        rows = n - 1234 * step
        idx = rng.choice(len(world), n, replace=False) if step else np.arange(n)
        app = world[idx]
        pts = rng.normal(0, 2, (n, 3)).astype(np.float32)
        d_p, d_a, d_n = ctx.to_device(pts), ctx.to_device(app), ctx.to_device(np.array([rows], np.int32))
        m.update_dev(d_p, d_a, n, d_n, m.history_dev)
        ctx.synchronize()
        dic.update(list(o32.transform_points(hist, pts[:rows])), list(app[:rows]))
        for d in (d_p, d_a, d_n):
            ctx.free(d)
        m.history_step_dev(d_X2)
        hist = P.iso_mul32(hist, P.iso_inv32(X2))
        assert np.array_equal(m.history(), hist)
    assert n < len(dic.pts) < 3 * n
    _same(m, dic.pts, dic.app)
    H = vo.synth.random_isometry(rng, 1.0, 1.0).astype(np.float32)
    m.transform(H)                                             # map = H * map (vo_complete.cpp:183)
    p, _ = m.read()
    assert p.tobytes() == o32.transform_points(H, np.array(dic.pts, np.float32)).tobytes()
    ctx.free(d_X1); ctx.free(d_X2)
    m.close()
