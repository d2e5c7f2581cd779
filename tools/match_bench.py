#!/usr/bin/env python3
"""Times the device-resident matcher alone on the 50k x 50k pair (events on the context's stream)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch
vo = g.load_package()
stream = torch.cuda.Stream()
ctx = vo.Context(0, stream.cuda_stream)
fp = vo.synth.frame_pair(int(os.environ.get("N", "50000")), seed=2000)
pipe = vo.FramePipeline(ctx, fp, n_iters=1)
for mode in (3, 2, 1):
    ctx.lib.vo_match_set_mode(ctx.h, mode)
    for _ in range(3): pipe.match()
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record(stream)
    for _ in range(reps): pipe.match()
    e1.record(stream); ctx.synchronize()
    print(f"mode {mode}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us, matches {pipe.counts()[0]}")

# batched form (frame = grid dimension): 200 frames per call, per variant
F = int(os.environ.get("F", "200"))
fps = [vo.synth.frame_pair(int(os.environ.get("N", "50000")), seed=4000 + (k % 8)) for k in range(8)]
bp = vo.BatchPipeline(ctx, [fps[k % 8] for k in range(F)], n_iters=1)
for mode in (3, 2):
    ctx.lib.vo_match_set_mode(ctx.h, mode)
    for _ in range(2): bp.run()
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(5): bp.run()
    e1.record(stream); ctx.synchronize()
    print(f"batched frames, matcher mode {mode}: {e0.elapsed_time(e1) / 5:.3f} ms per {F} frames (1 round), matches {bp.counts()[0][:3]}")
