#!/usr/bin/env python3
"""Experiment: 200 independent frames as one vo_frames_batch_dev call vs two half-batches on two contexts
(two HIP streams), so that the matcher of one half overlaps the solver of the other."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
N = int(os.environ.get("N", "50000")); F = int(os.environ.get("F", "200"))
fps = [vo.synth.frame_pair(N, seed=4000 + k) for k in range(8)]
def run(parts, reps=6):
    ctxs = [vo.Context(0) for _ in range(parts)]
    per = F // parts
    bps = [vo.BatchPipeline(c, [fps[(i * per + k) % 8] for k in range(per)], n_iters=50, with_appearance=True) for i, c in enumerate(ctxs)]
    for _ in range(2):
        for b in bps: b.run()
    for c in ctxs: c.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for b in bps: b.run()
    for c in ctxs: c.synchronize()
    dt = (time.perf_counter() - t0) / reps
    for b in bps: b.close()
    for c in ctxs: c.close()
    return dt
for parts in (1, 2, 4):
    dt = run(parts)
    print(f"{parts} stream(s): {dt*1e3:.3f} ms per {F} frames -> {F/dt:.0f} frames/s", flush=True)
