#!/usr/bin/env python3
"""Randomised campaign for the calls that take frames of DIFFERENT sizes (vo_match_appearances_batch_dev, vo_frames_batch_ragged_dev): random
numbers of frames, per-frame sizes from 0 to a few thousand points with either image the larger one, uniform / clustered / duplicated
appearances, matcher mode automatic / full scan / cell hash / exact-duplicate pass + cell hash, radii 0.03..0.3 -- every frame's pairs against the oracle; the whole loop body
(match -> join -> rounds in reference-order arithmetic -> triangulate) against the oracle frame by frame, bit for bit.
usage (GPU box): tools/fuzz_ragged.py [seed] [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
from oracle.oracle import Oracle, Camera as OCam
o32 = Oracle(32)
ctx = vo.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
n_match = n_frames_call = fails = 0


def appearance_sets(n, kind):
    """two sets sharing (a permutation of) n points, plus strays on one side"""
    if kind == 0: base = rng.uniform(-1, 1, (max(n, 1), 10))
    elif kind == 1: base = np.concatenate([rng.normal(0.3, 0.01, (max(n, 1) // 2 + 1, 10)), rng.uniform(-1, 1, (max(n, 1), 10))])[: max(n, 1)]
    else: base = rng.uniform(-1, 1, (max(n // 7, 1), 10))[rng.integers(0, max(n // 7, 1), max(n, 1))]
    x = base[:n].astype(np.float32)
    # a share of the copies displaced (no bitwise copy any more: the exact-duplicate pass leaves them open; few of them per frame
    # have the tree streamed past them, many go to the sorted search)
    share = float(rng.choice([0.0, 0.01, 0.04, 0.2, 1.0]))
    noise = rng.normal(0, 1e-3, (n, 10)) * (rng.uniform(0, 1, (n, 1)) < share)
    y = (x[rng.permutation(n)].astype(np.float64) + (0 if kind == 2 else noise)).astype(np.float32)
    extra = rng.uniform(-1, 1, (int(rng.integers(0, 1 + n // 5 + 3)), 10)).astype(np.float32)
    return (np.concatenate([x, extra]), y) if rng.integers(0, 2) else (x, np.concatenate([y, extra]))


while time.time() < t_end:
    F = int(rng.integers(1, 14))
    big = rng.integers(0, 5) == 0
    sizes = [int(rng.choice([0, 1, rng.integers(2, 60), rng.integers(60, 900), rng.integers(900, 6000 if big else 2500)])) for _ in range(F)]
    radius = float(rng.choice([0.1, 0.03, 0.3]))
    mode = int(rng.choice([0, 1, 3, 5]))       # 5: the exact-duplicate pass in front of the cell-hash search (what 0 picks from 8 large frames on)
    ctx.lib.vo_match_set_mode(ctx.h, mode)
    if rng.integers(0, 3) < 2:
        sets = [appearance_sets(n, int(rng.integers(0, 3))) for n in sizes]
        got = vo.match_batch_ragged(ctx, [s[0] for s in sets], [s[1] for s in sets], radius)
        for k, (a1, a2) in enumerate(sets):
            exp = o32.match(a1, a2, radius) if len(a1) and len(a2) else np.zeros((0, 2), np.int32)
            if not np.array_equal(got[k], exp):
                fails += 1; print("RAGGED MATCH FAIL", mode, radius, k, len(a1), len(a2), len(got[k]), len(exp))
        n_match += 1
    else:
        fps = []
        for n in sizes:
            n = max(n, 12)
            try:
                f = vo.synth.frame_pair(n, seed=int(rng.integers(0, 1 << 30)), drop=float(rng.choice([0.0, 0.1, 0.3])), distractors=int(rng.integers(0, 1 + n // 8)),
                                        model_drop=float(rng.choice([0.0, 0.2])))
            except Exception:
                continue
            if rng.integers(0, 2):                        # make the reference image the larger one
                m = max(1, int(0.7 * len(f["ref_app"])))
                f["cur_app"], f["cur_pts"] = f["cur_app"][:m].copy(), f["cur_pts"][:m].copy()
            fps.append(f)
        if not fps:
            continue
        rounds = int(rng.integers(0, 9)); thr = float(rng.choice([10000.0, 60.0])); keep = bool(rng.integers(0, 2))
        ctx.lib.vo_picp_batch_set_form(ctx.h, 3)          # reference-order arithmetic: bit for bit against the oracle
        cam = (fps[0]["rows"], fps[0]["cols"], fps[0]["z_near"], fps[0]["z_far"])
        res = vo.frames_batch_ragged(ctx, fps, fps[0]["K"], cam, n_iters=rounds, kernel_threshold=thr, radius=radius, keep_outliers=keep)
        ctx.lib.vo_picp_batch_set_form(ctx.h, 0)
        for k, (f, r) in enumerate(zip(fps, res)):
            m = o32.match(f["ref_app"], f["cur_app"], radius)
            j = o32.join(m, f["model_pairs"], linear=True)
            ok = np.array_equal(r["matches"], m) and np.array_equal(r["joined"], j)
            T = np.eye(4, dtype=np.float32)
            if rounds and len(j):
                T = o32.picp_solve_raw(OCam(*cam, f["K"], np.eye(4)), f["model"], f["cur_pts"], j, rounds, thr, keep)["T"][-1]
            ok = ok and np.array_equal(r["pose"], T, equal_nan=True)
            if ok and not np.isnan(T).any():
                e = o32.triangulate(f["K"], T, m, f["ref_pts"], f["cur_pts"])
                ok = np.array_equal(r["tri_pairs"], e[1]) and np.array_equal(r["tri_xyz"], e[0])
            if not ok:
                fails += 1; print("RAGGED FRAMES FAIL", mode, radius, rounds, thr, keep, k, len(f["ref_app"]), len(f["cur_app"]))
        n_frames_call += 1
ctx.lib.vo_match_set_mode(ctx.h, 0)
print("matcher calls", n_match, "frame calls", n_frames_call, "failures", fails)
