#!/usr/bin/env python3
"""One frame's matcher by size and mode (2 bucket-pruned scan, 3 cell-hash search, 1 full scan where it is the automatic choice):
where should the automatic mode switch?  usage: tools/match_sizes.py [n ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch
vo = g.load_package()
stream = torch.cuda.Stream()
ctx = vo.Context(0, stream.cuda_stream)
for n in [int(a) for a in sys.argv[1:]] or [2500, 5000, 10000, 20000, 35000, 50000, 100000, 200000]:
    fp = vo.synth.frame_pair(n, seed=2000)
    pipe = vo.FramePipeline(ctx, fp, n_iters=1)
    out = []
    for mode in (1, 2, 3, 0) if n <= 20000 else (2, 3, 0):
        ctx.lib.vo_match_set_mode(ctx.h, mode)
        for _ in range(3): pipe.match()
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40
        e0.record(stream)
        for _ in range(reps): pipe.match()
        e1.record(stream); ctx.synchronize()
        out.append(f"mode {mode}: {e0.elapsed_time(e1) / reps * 1e3:7.1f} us")
    print(f"n = {n:6d}: " + "   ".join(out), flush=True)
    pipe.close()
ctx.lib.vo_match_set_mode(ctx.h, 0)
