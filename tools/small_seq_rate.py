#!/usr/bin/env python3
"""Frames per second of the device-resident chain on frames of the reference dataset's size (config 5's shape: ~120 points in
view, 100 rounds per frame) -- where a frame is launches, not bytes.
usage (GPU box, repo root): [VO_HIP_LIB=...] tools/small_seq_rate.py [frames=121] [points=120] [rounds=100]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 121
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
R = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ctx = vo.Context(0)
seq = vo.synth.sequence(seed=3000, n_frames=F, n_visible=N)
for prematch in (False, True):
    sp = vo.SequencePipeline(ctx, seq, n_iters=R, prematch=prematch)
    sp.run(); ctx.synchronize()
    best = 1e9
    for _ in range(5):
        sp.start(); ctx.synchronize()
        t0 = time.perf_counter()
        for t in range(2, sp.F):
            sp.step(t)
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    traj = sp.trajectory()
    sp.close()
    print("%s: %d frames x ~%d points x %d rounds: %.1f us per frame (%.0f frames/s)%s  checksum %.9g" %
          (os.path.basename(vo.LIB_PATH), F, N, R, best / (F - 2) * 1e6, (F - 2) / best, "  [matched up front]" if prematch else "", float(np.abs(traj).sum())))
