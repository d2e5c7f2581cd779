"""Frames of DIFFERENT sizes in one call (vo_match_appearances_batch_dev, vo_frames_batch_ragged_dev): the reference's own
sequence has 14..127 points per frame (vo_complete.cpp:150-157).  Every frame of a ragged batch must come out exactly as
its own single-frame calls do -- roles (which set is the tree) included."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import vo_pipeline as vp

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(__file__), "golden", "example_data", "data")


def _frames():
    files, K, H, cam = vp._dataset(DATA)
    return [vp.read_meas(os.path.join(DATA, f)) for f in files], K, cam


def test_example_data_all_120_pairs_in_one_call(vo, ctx, o32):
    """BASELINE configs[4]'s matcher stage: the 120 consecutive frame pairs of the reference's dataset matched in ONE call;
    pairs equal to the per-frame calls, to the oracle, and (as sets) to the landmark-id overlap the data files carry."""
    fr, K, cam = _frames()
    assert len(fr) == 121
    a1 = [fr[k][1] for k in range(120)]; a2 = [fr[k + 1][1] for k in range(120)]
    sizes = {(len(x), len(y)) for x, y in zip(a1, a2)}
    assert len(sizes) > 60 and any(x > y for x, y in sizes) and any(x < y for x, y in sizes)      # really ragged, both roles
    got = vo.match_batch_ragged(ctx, a1, a2)
    assert len(got) == 120
    for k in range(120):
        single = vo.compute_correspondences_images(a1[k], a2[k], ctx=ctx)
        assert np.array_equal(got[k], single), k
        assert np.array_equal(got[k], o32.match(a1[k], a2[k])), k
        ids = vp.id_correspondences(fr[k][2], fr[k + 1][2])
        assert {tuple(p) for p in got[k].tolist()} == {tuple(p) for p in ids.tolist()}, k
    # the whole loop body for the same 120 pairs in one call: match + join against a per-frame model (here: a landmark per
    # reference point, some withheld), no round, triangulation from the identity pose -- every stage equal to its
    # single-frame call
    rng = np.random.default_rng(3)
    frames = []
    for k in range(120):
        n_ref = len(fr[k][0])
        keep = np.sort(rng.permutation(n_ref)[: max(1, int(0.8 * n_ref))])
        model = rng.uniform(-1, 1, (n_ref, 3)).astype(np.float32) + np.float32([0, 0, 3])
        frames.append(dict(ref_app=fr[k][1], cur_app=fr[k + 1][1], ref_pts=fr[k][0], cur_pts=fr[k + 1][0], model=model,
                           model_pairs=np.stack([keep, keep], 1).astype(np.int32)))
    res = vo.frames_batch_ragged(ctx, frames, K, cam, n_iters=0)
    for k, (f, r) in enumerate(zip(frames, res)):
        m = vo.compute_correspondences_images(f["ref_app"], f["cur_app"], ctx=ctx)
        assert np.array_equal(r["matches"], m), k
        assert np.array_equal(r["joined"], vo.extract_correspondences_world(m, f["model_pairs"], ctx=ctx)), k
        assert np.array_equal(r["pose"], np.eye(4, dtype=np.float32)), k
        xyz, pairs, _ = vo.triangulate_points(K, np.eye(4), m, f["ref_pts"], f["cur_pts"], ctx=ctx)
        assert np.array_equal(r["tri_pairs"], pairs) and np.array_equal(r["tri_xyz"], xyz), k


def test_ragged_synthetic_frames_equal_their_single_frame_calls(vo, ctx, o32):
    """random sizes 40..1500 with drops and distractors (so that either image may be the larger set), 8 rounds of the batched
    solver in its one-workgroup form: every frame of the ragged call == the same frame alone through vo_frames_batch_dev,
    bit for bit; matches / joins also == the oracle"""
    rng = np.random.default_rng(11)
    fps = [vo.synth.frame_pair(int(n), seed=600 + i, drop=0.15, distractors=int(n) // 10, model_drop=0.1)
           for i, n in enumerate(rng.integers(40, 1500, 14))]
    for i, f in enumerate(fps):                       # the generator's current image is always the larger one: cut every other one short
        if i % 2:
            m = int(0.8 * len(f["ref_app"]))
            f["cur_app"], f["cur_pts"] = f["cur_app"][:m].copy(), f["cur_pts"][:m].copy()
    assert any(len(f["ref_app"]) > len(f["cur_app"]) for f in fps) and any(len(f["ref_app"]) < len(f["cur_app"]) for f in fps)
    assert ctx.lib.vo_picp_batch_set_form(ctx.h, 2) == 0
    try:
        cam = (fps[0]["rows"], fps[0]["cols"], fps[0]["z_near"], fps[0]["z_far"])
        res = vo.frames_batch_ragged(ctx, fps, fps[0]["K"], cam, n_iters=8)
        for k, (f, r) in enumerate(zip(fps, res)):
            m = o32.match(f["ref_app"], f["cur_app"])
            assert np.array_equal(r["matches"], m), k
            assert np.array_equal(r["joined"], o32.join(m, f["model_pairs"])), k
            bp = vo.BatchPipeline(ctx, [f], n_iters=8)
            bp.run()
            assert np.array_equal(bp.fetch("match", 0), r["matches"]) and np.array_equal(bp.fetch("join", 0), r["joined"]), k
            assert np.array_equal(bp.poses()[0], r["pose"]), k
            assert np.array_equal(bp.stats()[0][:3], r["stats"][:3]), k
            assert np.array_equal(bp.fetch("tri_pairs", 0), r["tri_pairs"]) and np.array_equal(bp.fetch("tri_xyz", 0), r["tri_xyz"]), k
            bp.close()
            assert np.abs(r["pose"] - f["X_gt"]).max() < 5e-2, k
    finally:
        assert ctx.lib.vo_picp_batch_set_form(ctx.h, 0) == 0


def test_ragged_edge_cases(vo, ctx):
    """empty images, one point, equal sizes (the reference image is the tree on ties), sizes array of zeros"""
    rng = np.random.default_rng(5)
    base = rng.uniform(-1, 1, (50, 10)).astype(np.float32)
    a1 = [base[:0], base[:1], base[:20], base[:50], base[:7]]
    a2 = [base[:10], base[:1], base[:20][::-1].copy(), base[:0], base[3:30]]
    got = vo.match_batch_ragged(ctx, a1, a2)
    for k in range(5):
        assert np.array_equal(got[k], vo.compute_correspondences_images(a1[k], a2[k], ctx=ctx)), k
    assert len(got[0]) == 0 and len(got[3]) == 0 and got[1].tolist() == [[0, 0]]
    assert sorted(got[2].tolist()) == [[i, 19 - i] for i in range(20)]
    assert vo.match_batch_ragged(ctx, [], []) == []


def test_ragged_frames_through_the_cell_hash_search(vo, o32):
    """frames of different sizes AND roles through the sorted search (forced: mode 3; at these sizes the automatic choice is
    the full scan): per-frame sizes down to empty and one-point images next to frames of thousands of points, on uniform and
    clustered appearances -- every frame equal to the oracle"""
    c = vo.Context(0)
    assert c.lib.vo_match_set_mode(c.h, 3) == 0
    rng = np.random.default_rng(23)
    a1, a2 = [], []
    for k, n in enumerate([3000, 1, 0, 2500, 700, 4096, 1792, 1793, 5000]):
        base = rng.uniform(-1, 1, (max(n, 1), 10)).astype(np.float32)
        if k % 3 == 2:                                         # clustered: most points in a few cells
            base[: len(base) // 2] = (rng.normal(0.2, 0.02, (len(base) // 2, 10))).astype(np.float32)
        x = base[:n]
        perm = rng.permutation(n)
        y = (x[perm].astype(np.float64) + rng.normal(0, 2e-3, (n, 10))).astype(np.float32)
        extra = rng.uniform(-1, 1, (int(rng.integers(0, 400)), 10)).astype(np.float32)
        if k % 2:
            a1.append(np.concatenate([x, extra])); a2.append(y)       # the reference image is the larger set
        else:
            a1.append(x); a2.append(np.concatenate([y, extra]))
    a2[2] = rng.uniform(-1, 1, (40, 10)).astype(np.float32)            # empty against non-empty
    got = vo.match_batch_ragged(c, a1, a2)
    for k in range(len(a1)):
        exp = o32.match(a1[k], a2[k]) if len(a1[k]) and len(a2[k]) else np.zeros((0, 2), np.int32)
        assert np.array_equal(got[k], exp), (k, len(a1[k]), len(a2[k]), len(got[k]), len(exp))
    assert sum(len(g) for g in got) > 15000
    c.close()


def test_large_ragged_frames_pick_the_cell_hash_search_by_themselves(vo, ctx):
    """10 frames of 20k..26k points with different sizes: the automatic choice is the sorted search; pairs = the generator's
    permutation, and equal to every frame's own single-frame call"""
    fps = [vo.synth.frame_pair(20000 + 700 * k, seed=880 + k) for k in range(10)]
    got = vo.match_batch_ragged(ctx, [f["ref_app"] for f in fps], [f["cur_app"] for f in fps])
    for k, f in enumerate(fps):
        assert np.array_equal(got[k], f["gt_matches"]), k
        if k in (0, 9):
            assert np.array_equal(got[k], vo.compute_correspondences_images(f["ref_app"], f["cur_app"], ctx=ctx)), k


def test_ragged_argument_checks_and_clamping(vo, ctx):
    """the ragged entry points refuse inconsistent arguments before touching the stream, and per-frame sizes beyond the
    capacity / below zero are clamped instead of read past the arrays"""
    lib, h = ctx.lib, ctx.h
    rng = np.random.default_rng(9)
    a = rng.uniform(-1, 1, (3, 40, 10)).astype(np.float32)
    b = a[:, ::-1].copy()
    d_a, d_b = ctx.to_device(a), ctx.to_device(b)
    d_out, d_cnt = ctx.alloc(3 * 40 * 8), ctx.alloc(3 * 4)
    d_n = ctx.to_device(np.array([40, 10, 0], np.int32))
    NUL = C.c_void_p(0)
    call = lambda F, pa, c1, n1, pb, c2, n2, po, pc: lib.vo_match_appearances_batch_dev(
        h, C.c_int(F), pa, C.c_int(c1), n1, pb, C.c_int(c2), n2, C.c_float(0.1), po, pc)
    A, B, O, CN, N = (C.c_void_p(x) for x in (d_a, d_b, d_out, d_cnt, d_n))
    assert call(3, A, 40, N, B, 40, NUL, O, CN) < 0            # one size array without the other
    assert call(-1, A, 40, N, B, 40, N, O, CN) < 0
    assert call(70000, A, 40, N, B, 40, N, O, CN) < 0
    assert call(3, NUL, 40, N, B, 40, N, O, CN) < 0 and call(3, A, 40, N, B, 40, N, NUL, CN) < 0 and call(3, A, 40, N, B, 40, N, O, NUL) < 0
    assert call(0, NUL, 0, NUL, NUL, 0, NUL, NUL, CN) == 0      # nothing to do
    assert lib.vo_frames_batch_ragged_dev(h, NUL, NUL) < 0
    # sizes beyond the capacity and below zero: clamped to [0, cap]
    d_big = ctx.to_device(np.array([1000, -5, 40], np.int32))
    assert call(3, A, 40, C.c_void_p(d_big), B, 40, C.c_void_p(d_big), O, CN) == 0
    cnt = np.zeros(3, np.int32); ctx.d2h(cnt, d_cnt)
    assert cnt.tolist() == [40, 0, 40]
    out = np.zeros((3, 40, 2), np.int32); ctx.d2h(out, d_out)
    assert sorted(out[0].tolist()) == [[i, 39 - i] for i in range(40)]
    for d in (d_a, d_b, d_out, d_cnt, d_n, d_big):
        ctx.free(d)


def test_sizes_beyond_the_capacities_are_clamped_the_same_way_everywhere(vo, ctx, o32):
    """unequal capacities (cap1 = 40, cap2 = 90) and a per-frame size beyond its capacity (n1 = 120 > cap1): the search and
    the output compaction must clamp alike -- n1 becomes 40, so set 2 (n2 = 60) is the tree -- or the (ref, cur) columns of the
    frame come out swapped.  Full scan and, with larger sets, the sorted search."""
    import ctypes as C
    rng = np.random.default_rng(5)
    for cap1, cap2, sizes in ((40, 90, [(120, 60), (40, 90), (7, 3), (-5, 20)]), (2500, 5200, [(9000, 3000), (2500, 5200), (100, 2600), (2000, 1500)])):
        F = len(sizes)
        a1 = rng.uniform(-1, 1, (F, cap1, 10)).astype(np.float32); a2 = rng.uniform(-1, 1, (F, cap2, 10)).astype(np.float32)
        for f in range(F):                                   # copies, so that there is something to match
            k = min(cap1, cap2) // 2
            a2[f, :k] = a1[f, rng.permutation(cap1)[:k]]
        n1 = np.array([s[0] for s in sizes], np.int32); n2 = np.array([s[1] for s in sizes], np.int32)
        q = min(cap1, cap2)
        d = [ctx.to_device(x) for x in (a1, a2, n1, n2)]
        d_out, d_cnt = ctx.alloc(F * q * 8), ctx.alloc(F * 4)
        try:
            rc = ctx.lib.vo_match_appearances_batch_dev(ctx.h, C.c_int(F), C.c_void_p(d[0]), C.c_int(cap1), C.c_void_p(d[2]), C.c_void_p(d[1]),
                                                        C.c_int(cap2), C.c_void_p(d[3]), C.c_float(0.1), C.c_void_p(d_out), C.c_void_p(d_cnt))
            assert rc == 0, ctx.lib.vo_last_error()
            cnt = np.zeros(F, np.int32); ctx.d2h(cnt, d_cnt)
            out = np.zeros((F, q, 2), np.int32); ctx.d2h(out, d_out)
        finally:
            for x in d + [d_out, d_cnt]:
                ctx.free(x)
        for f in range(F):
            m1, m2 = int(np.clip(n1[f], 0, cap1)), int(np.clip(n2[f], 0, cap2))
            exp = o32.match(a1[f, :m1], a2[f, :m2])
            assert np.array_equal(out[f, : cnt[f]], exp), (cap1, cap2, f)
        assert cnt.sum() > 0
