#!/usr/bin/env python3
"""frames/s of vo_frames_batch_dev (all stages batched over F independent 50k-point frame pairs).
FORM=3 in the environment: the solver stage in reference-order arithmetic (vo_picp_batch_set_form)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch
vo = g.load_package()
n = int(os.environ.get("N", "50000"))
distinct = [vo.synth.frame_pair(n, seed=8000 + i) for i in range(4)]
stream = torch.cuda.Stream()
ctx = vo.Context(0, stream.cuda_stream)
form = int(os.environ.get("FORM", "0"))
assert ctx.lib.vo_picp_batch_set_form(ctx.h, form) == 0
for F in [int(a) for a in sys.argv[1:]] or [8, 32, 64, 200]:
    fps = [distinct[i % 4] for i in range(F)]
    bp = vo.BatchPipeline(ctx, fps, n_iters=50)
    bp.run(); ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get("REPS", "10"))
    e0.record(stream)
    for _ in range(reps): bp.run()
    e1.record(stream); ctx.synchronize()
    ms = e0.elapsed_time(e1) / reps
    P = bp.poses()
    err = max(float(np.abs(P[i] - fps[i]["X_gt"]).max()) for i in range(F))
    c = bp.counts()
    print(f"form {form} F={F}: {ms:.3f} ms per batch = {F / ms * 1e3:.0f} frames/s  ({ms / F * 1e3:.1f} us per frame)  pose err {err:.1e} counts {c[:, 0].tolist()}")
    bp.close()
