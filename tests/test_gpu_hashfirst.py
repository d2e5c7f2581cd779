"""The exact-duplicate pass of the matcher (match.hip "hash-first", modes 4 / 5 and the automatic mode from the sizes on
where it sorts): appearances are copied from frame to frame, so a query usually has a bitwise copy in the tree -- its nearest
neighbour at distance 0 -- which per-slice hash tables find without any search.  The pass must never change a result
(compute_correspondences_images, vo_complete.cpp:12-49; ties to the lowest index): every case here is held to the oracle
and to the plain searches (modes 1 / 3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctxs(vo, modes):
    out = []
    for m in modes:
        c = vo.Context(0)
        assert c.lib.vo_match_set_mode(c.h, m) == 0
        out.append(c)
    return out


def _check(vo, o32, ctxs, a, b, radius=0.1):
    exp = o32.match(a, b, radius)
    for c in ctxs:
        got = vo.compute_correspondences_images(a, b, radius=radius, ctx=c)
        assert np.array_equal(got, exp), (len(a), len(b))
    return exp


def test_hash_first_adversarial_rows(vo, o32):
    rng = np.random.default_rng(41)
    ctxs = _ctxs(vo, (1, 4, 5))
    base = rng.uniform(-1, 1, (3000, 10)).astype(np.float32)
    perm = rng.permutation(3000)
    # (1) plain copies, a permutation apart; with drops and strangers on both sides
    m = _check(vo, o32, ctxs, base, base[perm])
    assert len(m) == 3000
    strangers = rng.uniform(-1, 1, (400, 10)).astype(np.float32)
    _check(vo, o32, ctxs, np.concatenate([base[:2500], strangers]), np.concatenate([strangers[::-1][:100] * np.float32(0.5), base[perm][:2800]]))
    # (2) every tree row repeated five times, scattered: the LOWEST index of the copies must win
    tree = np.concatenate([base[:600]] * 5)[rng.permutation(3000)]
    m = _check(vo, o32, ctxs, tree, base[:600][::-1].copy())
    assert len(m) == 600
    # (3) one row, 3000 times (a chain of one entry, not of 3000), and a near copy of it in front
    same = np.repeat(base[:1], 3000, 0)
    same[0, 3] = np.nextafter(same[0, 3], np.float32(2))
    m = _check(vo, o32, ctxs, same, base[:1].repeat(40, 0))
    assert (m[:, 0] == 1).all()
    # (4) rows the pass must leave to the search: zeros, negative zeros, tiny values whose squared differences underflow to
    # zero (distance-0 ties between rows that are NOT copies), NaN, inf
    t = base[:2000].copy(); q = base[:2000][rng.permutation(2000)][:1500].copy()
    t[10:20, 2] = 0.0; t[20:30, 2] = -0.0
    q[:5] = t[10:15]; q[5:10] = t[20:25]; q[5:10, 2] = 0.0              # +0 against -0: equal, though not bitwise
    tiny = np.full((6, 10), 1e-30, np.float32)
    tiny[2, 0] = 1.5e-30; tiny[4, 5] = 0.5e-30                          # rows 0,1,3,5 identical; 2 and 4 differ by less than sqrt(min float)
    t[100:106] = tiny; q[20:23] = tiny[[5, 2, 4]]
    t[200, 4] = np.nan; q[30] = t[200]; t[201, 7] = np.inf; q[31] = t[201]; q[32, 0] = -np.inf
    for a, b in ((t, q), (q, t)):
        _check(vo, o32, ctxs, a, b)
    # (5) radius edge: nothing can match at radius 0 (0 < 0 is false), everything safe at a tiny positive radius
    assert len(_check(vo, o32, ctxs, base[:2100], base[:2100][::-1].copy(), radius=0.0)) == 0
    assert len(_check(vo, o32, ctxs, base[:2100], base[:2100][::-1].copy(), radius=1e-6)) == 2100
    # (6) values around the bound that separates the rows the pass takes from those it leaves (2^-40)
    edge = base[:2400].copy()
    edge[::3, 1] = np.float32(2.0 ** -40); edge[1::3, 1] = np.nextafter(np.float32(2.0 ** -40), np.float32(0))
    _check(vo, o32, ctxs, edge, edge[rng.permutation(2400)][:2200].copy())
    for c in ctxs: c.close()


def test_hash_first_many_slices_and_batches(vo, ctx, o32):
    """several parts per table (70k points: eight parts), frames of different sizes in one call with either
    image the larger one, and a batch whose frames differ in how many queries the pass leaves open (none ... all)"""
    rng = np.random.default_rng(43)
    big = rng.uniform(-1, 1, (70000, 10)).astype(np.float32)
    q = big[rng.permutation(70000)][:60000].copy()
    q[:3000] += np.float32(0.01)                                        # near copies: the search must find them
    q[3000:4000] = rng.uniform(-1, 1, (1000, 10)).astype(np.float32)    # strangers
    c1, c5 = _ctxs(vo, (3, 5))
    exp = vo.compute_correspondences_images(big, q, ctx=c1)
    assert 58000 < len(exp) <= 60000
    assert np.array_equal(vo.compute_correspondences_images(big, q, ctx=c5), exp)
    assert np.array_equal(vo.compute_correspondences_images(big, q, ctx=ctx), exp)          # automatic mode
    assert np.array_equal(vo.compute_correspondences_images(q, big, ctx=ctx), exp[:, ::-1])
    # ragged batch, 12 frames, open queries from 0 % to 100 %
    a1, a2 = [], []
    for k in range(12):
        n = int(rng.integers(2500, 9000))
        t = rng.uniform(-1, 1, (n, 10)).astype(np.float32)
        m = int(n * rng.uniform(0.5, 1.0))
        qq = t[rng.permutation(n)][:m].copy()
        n_open = int(m * k / 11)
        qq[:n_open] += rng.normal(0, 0.02, (n_open, 10)).astype(np.float32)
        if k % 2: a1.append(t); a2.append(qq)
        else: a1.append(qq); a2.append(t)
    for c in (c5, ctx):
        got = vo.match_batch_ragged(c, a1, a2)
        for k in range(12):
            assert np.array_equal(got[k], o32.match(a1[k], a2[k])), k
    c1.close(); c5.close()


def test_hash_first_in_batched_calls_of_equal_frames(vo, o32):
    """vo_match_appearances_batch_dev on frames of ONE size (the non-ragged path: blockIdx.z frames in the bucket-pruned
    kernels, XCD-mapped frames in the cell-hash ones) with the exact-duplicate pass in front: 3 and 9 frames per call, frames
    whose share of open queries runs from none to all -- every frame's pairs equal to the oracle's, modes 4 and 5."""
    rng = np.random.default_rng(47)
    n = 2600
    for F in (3, 9):
        fps = []
        for k in range(F):
            f = vo.synth.frame_pair(n, seed=900 + k)
            a = f["cur_app"].copy()
            n_open = int(len(a) * k / max(F - 1, 1))
            idx = rng.permutation(len(a))[:n_open]
            a[idx] += rng.normal(0, 0.01, (n_open, 10)).astype(np.float32)       # near copies: only the search finds them
            f = dict(f); f["cur_app"] = a
            fps.append(f)
        for mode in (4, 5):
            c = vo.Context(0)
            assert c.lib.vo_match_set_mode(c.h, mode) == 0
            bp = vo.BatchPipeline(c, fps, n_iters=1)
            bp.match_only()
            cnt = bp.counts()[0]
            for k in range(F):
                exp = o32.match(fps[k]["ref_app"], fps[k]["cur_app"])
                assert cnt[k] == len(exp) and np.array_equal(bp.fetch("match", k), exp), (F, mode, k)
            bp.close(); c.close()


def _row_hash(rows):
    """match.hip: row_hash, restated (uint32 arithmetic) -- white-box: used only to BUILD inputs that overflow a part"""
    w = np.ascontiguousarray(rows, np.float32).view(np.uint32).astype(np.uint64)
    M = np.uint64(0xffffffff)
    rotl = lambda x, r: ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & M
    x = w[:, 0]
    for k in range(1, 10):
        x = (rotl(x, 7) + w[:, k]) & M if k & 1 else rotl(x, 11) ^ w[:, k]
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x2c1b3c6d)) & M
    x ^= x >> np.uint64(12); x = (x * np.uint64(0x297a2d39)) & M
    x ^= x >> np.uint64(15)
    return x.astype(np.uint32)


def test_hash_first_part_overflow_falls_back_to_the_search(vo, o32):
    """A tree whose rows all hash into ONE part of the table (picked with the numpy restatement of the hash): the part's queue
    overflows, the part is stored empty and every query goes to the search -- among them copies that exist twice in the
    tree, whose LOWER index must come back (a table that had taken only some of the rows could hold the higher one)."""
    rng = np.random.default_rng(53)
    pool = rng.uniform(-1, 1, (120000, 10)).astype(np.float32)
    h = _row_hash(pool)
    # 17 000 tree points: hash_plan gives 2^14-word parts, 4 of them (log2p = 2, average share 6400, queue 9600)
    part0 = pool[(h >> np.uint32(30)) == 0]
    assert len(part0) > 17000
    tree = part0[:17000].copy()
    tree[9000:9200] = tree[100:300]                              # 200 rows twice: the copy at the lower index must win
    q = np.concatenate([tree[100:300], tree[5000:6000], rng.uniform(-1, 1, (300, 10)).astype(np.float32)])
    q = q[rng.permutation(len(q))]
    exp = o32.match(tree, q)
    assert len(exp) == 1200 and set(exp[:, 0]) >= set(range(100, 300)) and not (set(exp[:, 0]) & set(range(9000, 9200)))
    for mode in (4, 5):
        c = vo.Context(0)
        assert c.lib.vo_match_set_mode(c.h, mode) == 0
        for _ in range(3):                                       # (the queues' record order varies from run to run)
            assert np.array_equal(vo.compute_correspondences_images(tree, q, ctx=c), exp), mode
        c.close()
    # the same rows spread over all parts (no overflow): the pass answers them itself, same pairs
    mixed = pool[:17000].copy(); mixed[9000:9200] = mixed[100:300]
    q2 = np.concatenate([mixed[100:300], mixed[5000:6000]])
    c = vo.Context(0)
    assert c.lib.vo_match_set_mode(c.h, 5) == 0
    assert np.array_equal(vo.compute_correspondences_images(mixed, q2, ctx=c), o32.match(mixed, q2))
    c.close()


def test_few_open_queries_have_the_tree_streamed_past_them(vo, o32):
    """Mode 5 with a FEW queries left open by the exact-duplicate pass (match.hip: open_collect_kernel / open_scan_kernel: the
    open queries ordered by cell, the tree streamed once past them) -- counts on both sides of the route's limits (nq / 16,
    2560), open queries that tie between several tree points (lowest index), that sit exactly at the radius (strictly inside
    only), whose nearest neighbour is a row the pass must not hash (zeros), that crowd into one cell (the route declines),
    that come at the END of the query array (new landmarks appended), NaN / inf rows; single calls, both roles, and a
    ragged batch.  Every result equal to the oracle's and to the plain search's (mode 3)."""
    rng = np.random.default_rng(59)
    c3, c5 = _ctxs(vo, (3, 5))
    R = np.float32(0.1)

    def both(a, b, radius=0.1, oracle=True):
        # the plain search (mode 3) is held to the oracle; the pass (mode 5) to the plain search -- the largest cases skip the
        # oracle's brute force (seconds each on the CPU)
        exp = vo.compute_correspondences_images(a, b, radius=radius, ctx=c3)
        if oracle: assert np.array_equal(exp, o32.match(a, b, radius))
        assert np.array_equal(vo.compute_correspondences_images(a, b, radius=radius, ctx=c5), exp)
        assert np.array_equal(vo.compute_correspondences_images(b, a, radius=radius, ctx=c5), exp[:, ::-1])
        return exp

    n = 48000
    tree = rng.uniform(-1, 1, (n, 10)).astype(np.float32)
    q0 = tree[rng.permutation(n)][:44000].copy()
    for n_open in (1, 7, 300, 2559, 2560, 2561, 2749, 2750, 2751, 6000):      # 44000 / 16 = 2750
        q = q0.copy()
        idx = rng.permutation(len(q))[:n_open]
        q[idx] += rng.normal(0, 0.01, (n_open, 10)).astype(np.float32)        # near copies: most still inside the radius
        q[idx[: n_open // 3]] = rng.uniform(-1, 1, (n_open // 3, 10)).astype(np.float32)   # strangers: most have no match
        exp = both(tree, q, oracle=n_open in (7, 2560, 2751))
        assert len(exp) >= len(q) - n_open
    # new landmarks appended at the end of the query array (whole lookup workgroups of open queries)
    q = np.concatenate([q0[:30000], tree[:1500] + np.float32(0.004)]).astype(np.float32)
    assert len(both(tree, q, oracle=False)) > 31000
    # ties: an open query in the middle of two tree points at bitwise equal distance (copies of one row, displaced): lowest index
    t2 = tree.copy()
    t2[40000:40050] = t2[100:150]                                             # the same rows twice
    q = q0[:12000].copy()
    q[:50] = t2[100:150]; q[:50, 0] += np.float32(0.03)                       # 0.03 from both copies
    exp = both(t2, q)
    hit = {int(b): int(a) for a, b in exp}
    assert all(hit[k] == 100 + k for k in range(50))
    # exactly at the radius: with radius 0.25 and one component displaced by 0.25 the squared distance equals radius^2 bit for
    # bit (both are the float 0.0625): no match; one ulp closer: a match
    t3 = tree[:16000].copy(); t3[:, 0] = np.float32(0.5) * np.sign(t3[:, 0]) * np.abs(t3[:, 0])      # keep x +- 0.25 exact-ish
    t3[:200, 0] = np.float32(0.25)
    q = t3[rng.permutation(16000)][:14000].copy()
    q[:100] = t3[:100]; q[:100, 0] = np.float32(0.5)                          # d = 0.25 exactly
    q[100:200] = t3[100:200]; q[100:200, 0] = np.nextafter(np.float32(0.5), np.float32(0))
    exp = both(t3, q, radius=0.25)
    hit = {int(b): int(a) for a, b in exp}
    assert not any(k in hit for k in range(100)) and all(hit[k] == k for k in range(100, 200))
    # rows the pass leaves to the search although they HAVE a copy (a zero component), NaN and inf rows, among few open queries
    t4 = tree[:20000].copy(); t4[50:90, 6] = 0.0; t4[300, 2] = np.nan; t4[301, 3] = np.inf
    q = t4[rng.permutation(20000)][:18000].copy()
    q[:40] = t4[50:90]; q[40] = t4[300]; q[41] = t4[301]; q[42, 9] = -np.inf
    exp = both(t4, q)
    hit = {int(b): int(a) for a, b in exp}
    assert all(hit[k] == 50 + k for k in range(40)) and 40 not in hit and 41 not in hit and 42 not in hit
    # crowded: 1500 open queries within one cell's width of one another (the route declines: the sorted search takes them)
    q = q0[:30000].copy()
    q[:1500] = (tree[7] + rng.uniform(-0.02, 0.02, (1500, 10))).astype(np.float32)
    both(tree, q, oracle=False)
    # a ragged batch in mode 5: frames with 0, 1, a few, many open queries, either image the larger one
    a1, a2 = [], []
    for k, n_open in enumerate((0, 1, 40, 500, 900, 0, 9, 3000, 150, 0)):
        nk = int(rng.integers(9000, 16000))
        t = rng.uniform(-1, 1, (nk, 10)).astype(np.float32)
        qq = t[rng.permutation(nk)][: int(nk * 0.9)].copy()
        qq[:n_open] += rng.normal(0, 0.015, (n_open, 10)).astype(np.float32)
        if k % 2: a1.append(t); a2.append(qq)
        else: a1.append(qq); a2.append(t)
    got = vo.match_batch_ragged(c5, a1, a2)
    for k in range(len(a1)):
        assert np.array_equal(got[k], o32.match(a1[k], a2[k])), k
    c3.close(); c5.close()


def test_frames_without_copies_skip_the_pass(vo, o32):
    """hash_rows_kernel looks for eight sampled queries of a frame (positions k * (nq >> 3)) while the tree streams past; a
    frame in which none is found builds no table and looks nothing up.  Whatever it decides, the pairs are the search's: data
    without any copy; copies only at positions the sample does not look at (the pass is skipped although it would have found
    them); a copy only at a sampled position; fewer than eight queries; the same in one ragged batch, either image the tree."""
    rng = np.random.default_rng(61)
    c3, c5 = _ctxs(vo, (3, 5))
    n = 6000
    tree = rng.uniform(-1, 1, (n, 10)).astype(np.float32)
    near = (tree[rng.permutation(n)][:5200] + rng.normal(0, 0.004, (5200, 10))).astype(np.float32)     # no bitwise copy anywhere
    sampled = np.arange(8) * (5200 >> 3)
    only_unsampled = near.copy()
    rest = np.setdiff1d(np.arange(5200), sampled)
    only_unsampled[rest[:3000]] = tree[rng.permutation(n)[:3000]]                                      # copies, none at a sampled position
    only_sampled = near.copy(); only_sampled[sampled[3]] = tree[77]
    tiny = tree[[5, 9, 11, 200, 4000]].copy(); tiny[1] += np.float32(0.003)
    cases = [(tree, near), (tree, only_unsampled), (tree, only_sampled), (tree, tiny), (tree[:40], tree[:40][::-1].copy())]
    for a, b in cases:
        exp = o32.match(a, b)
        for c in (c3, c5):
            assert np.array_equal(vo.compute_correspondences_images(a, b, ctx=c), exp)
            assert np.array_equal(vo.compute_correspondences_images(b, a, ctx=c), o32.match(b, a))
    a1 = [c[k % 2] for k, c in enumerate(cases)]; a2 = [c[1 - k % 2] for k, c in enumerate(cases)]
    got = vo.match_batch_ragged(c5, a1, a2)
    for k in range(len(cases)):
        assert np.array_equal(got[k], o32.match(a1[k], a2[k])), k
    c3.close(); c5.close()
