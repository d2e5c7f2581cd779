#!/usr/bin/env python3
"""One 50k-point frame (match + join + transform + 50 rounds + triangulate), device-resident: its ~20 launches one by one
against the whole frame captured into a hipGraph (FramePipeline.capture_frame) and replayed."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch
vo = g.load_package()
stream = torch.cuda.Stream()
ctx = vo.Context(0, stream.cuda_stream)
for n in [int(a) for a in sys.argv[1:]] or [127, 2000, 50000]:
    fp = vo.synth.frame_pair(n, seed=2000)
    pipe = vo.FramePipeline(ctx, fp, n_iters=50)
    def timed(fn, reps=50):
        for _ in range(3): fn()
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps): fn()
        e1.record(stream); ctx.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    plain = timed(pipe.frame)
    T0 = pipe.pose().copy(); c0 = pipe.counts().copy()
    pipe.capture_frame()
    graph = timed(pipe.frame_graph)
    assert np.array_equal(pipe.pose(), T0) and np.array_equal(pipe.counts(), c0)
    print(f"n={n}: launch by launch {plain:.1f} us per frame, one graph {graph:.1f} us per frame", flush=True)
    pipe.close()
