#!/usr/bin/env python3
"""Where does a PICP round spend its time?  Uses the diagnostic build (make -C visual-odometry_amd/csrc stamps), whose
round kernel records s_memtime (shader clock) at phase boundaries of workgroup 0 and s_memrealtime (the constant 100 MHz
reference clock) at its first instruction -- the latter calibrates the former's tick, per step -- on
the headline geometry (one 50k pair, 50 rounds per step, graph replay: what bench.py times).  Per step the event time of
the same launches is taken too, so that the phase sum can be held against launch_us of the SAME run.
usage (GPU box): VO_HIP_LIB=$PWD/visual-odometry_amd/libvo_hip_stamps.so python3 tools/stamp_rounds.py [steps=30] [out.json]
(oneRound: /root/reference/src/picp_solver.cpp:98-112)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
import torch  # noqa: E402

vo = g.load_package()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
out_path = sys.argv[2] if len(sys.argv) > 2 else None
stream = torch.cuda.Stream()
ctx = vo.Context(0, stream.cuda_stream)
assert hasattr(ctx.lib, "vo_debug_get_stamps"), "needs the stamps build: VO_HIP_LIB=.../libvo_hip_stamps.so"
fp = vo.synth.frame_pair(50000, seed=2000)
ITERS = 50
pipe = vo.FramePipeline(ctx, fp, n_iters=ITERS, kernel_threshold=10000.0)
names = ["partial rows of the previous launch arrive (all loads issued at the first instruction)",
         "rows staged through LDS, barrier, this lane's entry of H / b summed",
         "tail: 6x6 LDL^T, sin/cos, pose composition",
         "barrier, linearisation of this workgroup's 256 correspondences",
         "workgroup reduction (quad DPP + LDS, 2 barriers)",
         "partial row stored"]
rows, ev_us = [], []
with torch.cuda.stream(stream):
    pipe.match(); pipe.join(); pipe.transform()
    for _ in range(3): pipe.picp()
    ctx.synchronize()
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); pipe.picp(); e1.record(stream)
        ctx.synchronize()
        ev_us.append(e0.elapsed_time(e1) * 1e3 / ITERS)
        st = np.zeros((128, 8), np.uint64)
        assert ctx.lib.vo_debug_get_stamps(pipe.solver, st.ctypes.data_as(C.c_void_p)) == 0
        rows.append(st[1:ITERS].astype(np.int64))          # rounds 1..49 run picp_round_kernel<true,false,...>
# s_memtime ticks per microsecond, per step: first to last round's first instruction on both clocks (100 ticks of the
# reference clock per microsecond)
tpu = np.array([(r[-1, 0] - r[0, 0]) / ((r[-1, 7] - r[0, 7]) / 100.0) for r in rows])
st = np.concatenate(rows)                                   # (steps * 49, 8)
tick_us = 1.0 / float(np.median(tpu))
d = np.diff(st[:, :7], axis=1) * tick_us
in_kernel = (st[:, 6] - st[:, 0]) * tick_us
r2r = np.concatenate([np.diff(r[:, 0]) for r in rows]) * tick_us      # stamp 0 -> next round's stamp 0 (within a step)
launch_us = float(np.median(ev_us))
rep = {"what": "s_memtime stamps of workgroup 0 of picp_round_kernel<true,false,true,false>, VO_STAMPS build, 50k correspondences, "
               "196 workgroups, graph replay of 50 rounds per step",
       "rounds": int(len(st)), "steps": steps, "shader_clock_ticks_per_us": float(np.median(tpu)),
       "clock_note": "s_memtime ticks converted with the step's own ratio to s_memrealtime (100 MHz)",
       "phases_us_mean": {n: float(d[:, k].mean()) for k, n in enumerate(names)},
       "phases_us_median": {n: float(np.median(d[:, k])) for k, n in enumerate(names)},
       "in_kernel_us_mean": float(in_kernel.mean()), "in_kernel_us_median": float(np.median(in_kernel)),
       "round_to_round_us_mean": float(r2r.mean()), "round_to_round_us_median": float(np.median(r2r)),
       "kernel_boundary_us_mean": float(r2r.mean() - in_kernel.mean()),
       "launch_us_same_run_events": launch_us,
       "phase_sum_plus_boundary_vs_launch_us": float(r2r.mean() / launch_us)}
print("phase (workgroup 0)                                                                        mean us  median us")
for k, n in enumerate(names):
    print(f"{n:90s}{d[:, k].mean():8.3f}{np.median(d[:, k]):10.3f}")
print(f"{'in-kernel total (first instruction -> partial stored)':90s}{in_kernel.mean():8.3f}{np.median(in_kernel):10.3f}")
print(f"{'kernel boundary (partial stored -> first instruction of the next launch)':90s}{r2r.mean() - in_kernel.mean():8.3f}")
print(f"{'round to round (first instruction -> first instruction of the next launch)':90s}{r2r.mean():8.3f}{np.median(r2r):10.3f}")
print(f"event time of the same launches / 50 (launch_us of this run, stamps build): {launch_us:.3f} us; "
      f"round to round / launch_us = {r2r.mean() / launch_us:.3f}  ({len(st)} rounds)")
if out_path:
    with open(out_path, "w") as f:
        json.dump(rep, f, indent=1)
