#!/usr/bin/env python3
"""Probe: F independent frames as chunks on several contexts (own stream and workspaces each) against one call.
The solver stage is VALU-bound on one CU per problem, the other stages are memory-bound: do they overlap?
  overlap_probe.py 1 2 4          equal chunks on that many streams, all started together
  STAGGER_US=400 overlap_probe.py 2   stream k starts k x STAGGER_US late (a spinning one-wave kernel in front of it)
  CHUNKS=512,512,512,64 STREAMS=2 overlap_probe.py   explicit chunk sizes dealt round-robin to the streams"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch
vo = g.load_package()
n = int(os.environ.get("N", "50000"))
F = int(os.environ.get("F", "200"))
reps = int(os.environ.get("REPS", "10"))
stagger = float(os.environ.get("STAGGER_US", "0"))
distinct = [vo.synth.frame_pair(n, seed=8000 + i) for i in range(4)]
CLK = 100e6   # torch.cuda._sleep counts the constant 100 MHz clock on this stack (calibrated below)

def calibrate():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000); e0.record(); torch.cuda._sleep(1000000); e1.record()
    torch.cuda.synchronize()
    return 1000000 / (e0.elapsed_time(e1) * 1e-3)
CLK = calibrate()
print(f"_sleep clock: {CLK / 1e6:.1f} MHz")

def run(chunks, n_streams, label):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    ctxs = [vo.Context(0, s.cuda_stream) for s in streams]
    lo = 0; bps = []
    for k, c in enumerate(chunks):
        bps.append((k % n_streams, vo.BatchPipeline(ctxs[k % n_streams], [distinct[i % 4] for i in range(lo, lo + c)], n_iters=50)))
        lo += c
    for _, bp in bps: bp.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if stagger > 0:
            for k in range(1, n_streams):
                with torch.cuda.stream(streams[k]): torch.cuda._sleep(int(k * stagger * 1e-6 * CLK))
        for _, bp in bps: bp.run()
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    tot = sum(chunks)
    print(f"{label}: {ms:.3f} ms per {tot} frames = {tot / ms * 1e3:.0f} frames/s ({ms / tot * 1e3:.2f} us per frame)", flush=True)
    for _, bp in bps: bp.close()

if os.environ.get("CHUNKS"):
    ch = [int(x) for x in os.environ["CHUNKS"].split(",")]
    ns = int(os.environ.get("STREAMS", "2"))
    run(ch, ns, f"chunks {ch} on {ns} stream(s), stagger {stagger:.0f} us")
else:
    for C in [int(a) for a in sys.argv[1:]] or [1, 2]:
        bounds = [F * k // C for k in range(C + 1)]
        run([bounds[k + 1] - bounds[k] for k in range(C)], C, f"{C} equal chunk(s), stagger {stagger:.0f} us")
