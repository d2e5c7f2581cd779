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
for mode in (2, 1):
    ctx.lib.vo_match_set_mode(ctx.h, mode)
    for _ in range(3): pipe.match()
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record(stream)
    for _ in range(reps): pipe.match()
    e1.record(stream); ctx.synchronize()
    print(f"mode {mode}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us, matches {pipe.counts()[0]}")
