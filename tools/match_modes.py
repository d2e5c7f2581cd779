#!/usr/bin/env python3
"""Matcher stage alone, per mode: 200 x 50k frames per call (vo_match_appearances_batch_dev) and one 50k frame
(vo_match_appearances_dev), on frames whose queries all have a copy in the tree and on frames where a share of them
(OPEN, default 0.1) was perturbed so that the exact-duplicate pass leaves them to the search.
Modes: 2 bucket-pruned, 3 cell-hash, 4 / 5 the same behind the exact-duplicate pass, 0 automatic.
OPENS=0,0.01,0.1,0.5,1 : the batched call only, modes 3 and 5, one line per share of open queries in EVERY frame."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch
vo = g.load_package()
stream = torch.cuda.Stream()
ctx = vo.Context(0, stream.cuda_stream)
N = int(os.environ.get("N", "50000"))
F = int(os.environ.get("F", "200"))
OPEN = float(os.environ.get("OPEN", "0.1"))


def perturbed(fp, share, seed):
    f = dict(fp)
    rng = np.random.default_rng(seed)
    a = f["cur_app"].copy()
    k = int(share * len(a))
    idx = rng.permutation(len(a))[:k]
    a[idx] += rng.normal(0, 0.005, (k, 10)).astype(np.float32)      # still within the radius of its landmark: a match the search must find
    f["cur_app"] = a
    return f


def timed(fn, reps):
    for _ in range(2): fn()
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): fn()
    e1.record(stream); ctx.synchronize()
    return e0.elapsed_time(e1) / reps


base = [vo.synth.frame_pair(N, seed=4000 + k) for k in range(8)]
if os.environ.get("OPENS"):
    for share in [float(x) for x in os.environ["OPENS"].split(",")]:
        fps = base if share == 0 else [perturbed(f, share, 70 + i) for i, f in enumerate(base)]
        bp = vo.BatchPipeline(ctx, [fps[k % 8] for k in range(F)], n_iters=1)
        row = []
        ref = None
        for mode in (3, 5):
            ctx.lib.vo_match_set_mode(ctx.h, mode)
            ms = timed(bp.match_only, 5)
            got = (bp.counts()[0].copy(), bp.fetch("match", 0).copy())
            if ref is None: ref = got
            row.append((mode, ms, np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])))
        print(f"batched {F} x {N}, {share:.0%} of every frame's queries without a copy: " +
              ", ".join(f"mode {m}: {ms:.3f} ms{'' if same else ' DIFFERENT'}" for m, ms, same in row) + f"; matches {ref[0][:2].tolist()}", flush=True)
        bp.close()
    sys.exit(0)
QUICK = os.environ.get("QUICK", "0") == "1"          # only the frames whose queries all have a copy
for label, fps in (("all copies", base), (f"{OPEN:.0%} open", [perturbed(f, OPEN, 70 + i) for i, f in enumerate(base)]))[:1 if QUICK else 2]:
    bp = vo.BatchPipeline(ctx, [fps[k % 8] for k in range(F)], n_iters=1)
    ref = None
    for mode in (3, 5, 0):
        ctx.lib.vo_match_set_mode(ctx.h, mode)
        ms = timed(bp.match_only, 5)
        c = bp.counts()[0]
        m0 = bp.fetch("match", 0)
        if ref is None: ref = (c.copy(), m0.copy())
        same = np.array_equal(c, ref[0]) and np.array_equal(m0, ref[1])
        print(f"batched {F} x {N} [{label}] mode {mode}: {ms:.3f} ms per call, matches {c[:3].tolist()} same-as-mode-3 {same}", flush=True)
    bp.close()
    pipe = vo.FramePipeline(ctx, fps[0], n_iters=1)
    for mode in (2, 4, 3, 5, 0):
        ctx.lib.vo_match_set_mode(ctx.h, mode)
        us = timed(pipe.match, 30) * 1e3
        print(f"one frame {N} [{label}] mode {mode}: {us:.1f} us, matches {pipe.counts()[0]}", flush=True)
    pipe.close()
ctx.lib.vo_match_set_mode(ctx.h, 0)
