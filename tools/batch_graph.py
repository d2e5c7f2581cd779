#!/usr/bin/env python3
"""One vo_frames_batch_dev call (200 x 50k frames) launched kernel by kernel against the same call captured once into a
hipGraph (vo_ctx_begin_capture / vo_ctx_end_capture) and replayed with one launch: what the ~35 kernel boundaries cost."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch
vo = g.load_package()
n = int(os.environ.get("N", "50000")); F = int(os.environ.get("F", "200")); reps = int(os.environ.get("REPS", "10"))
distinct = [vo.synth.frame_pair(n, seed=8000 + i) for i in range(4)]
stream = torch.cuda.Stream()
ctx = vo.Context(0, stream.cuda_stream)
bp = vo.BatchPipeline(ctx, [distinct[i % 4] for i in range(F)], n_iters=50)
bp.run(); ctx.synchronize()
def timed(fn):
    fn(); ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps): fn()
    e1.record(stream); ctx.synchronize()
    return e0.elapsed_time(e1) / reps
plain = timed(bp.run)
P0 = bp.poses().copy(); c0 = bp.counts().copy()
gr = C.c_void_p()
assert ctx.lib.vo_ctx_begin_capture(ctx.h) == 0
try:
    bp.run()
finally:
    rc = ctx.lib.vo_ctx_end_capture(ctx.h, C.byref(gr))
assert rc == 0, ctx.lib.vo_last_error()
replay = timed(lambda: ctx.lib.vo_graph_launch(gr))
assert np.array_equal(bp.poses(), P0) and np.array_equal(bp.counts(), c0)
print(f"F={F}: kernel by kernel {plain:.3f} ms, graph replay {replay:.3f} ms ({(plain - replay) * 1e3:.0f} us), same poses and counts")
ctx.lib.vo_graph_destroy(gr)
bp.close()
