#!/usr/bin/env python3
"""Where does a PICP round spend its time?  Uses the diagnostic build (make -C visual-odometry_amd/csrc stamps)
whose round kernel records s_memtime (100 MHz constant clock on gfx950 -> 10 ns ticks) at phase boundaries
of workgroup 0.  Never quote this build's run time; read the shares.
usage (GPU box): VO_HIP_LIB=visual-odometry_amd/libvo_hip_stamps.so python tools/stamp_rounds.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

vo = g.load_package()
ctx = vo.Context(0)
fp = vo.synth.frame_pair(50000, seed=2000)
mp = dict(fp["model_pairs"].tolist())
corr = np.array([(c, mp[r]) for r, c in fp["gt_matches"].tolist()], np.int32)
cam = vo.Camera(480, 640, 0, 10, fp["K"], np.eye(4), ctx=ctx)
s = vo.PICPSolver(ctx)
s.setKernelThreshold(10000.0)
s.init(cam, fp["model"], fp["cur_pts"])
for _ in range(3):
    s.init(cam, fp["model"], fp["cur_pts"])
    s.solve(corr, False, 50)
    s.numInliers()
st = np.zeros((128, 8), np.uint64)
assert ctx.lib.vo_debug_get_stamps(s.h, st.ctypes.data_as(C.c_void_p)) == 0
st = st[1:50].astype(np.int64)            # rounds 1..49 run the <true,false> kernel
names = ["load partials (+issue)", "LDS sum + expand + 2 barriers", "tail: pivot order, LDLT, sincos, pose", "barrier + linearise", "block reduce (DPP+LDS)", "store partial"]
d = np.diff(st[:, :7], axis=1)
tick_ns = 10.0
print("phase                                     median ns")
for k, nm in enumerate(names):
    print(f"{nm:42s}{np.median(d[:, k]) * tick_ns:8.0f}")
print(f"{'in-kernel total (stamp 0 -> 6)':42s}{np.median(st[:, 6] - st[:, 0]) * tick_ns:8.0f}")
print(f"{'round to round (stamp 0 -> next stamp 0)':42s}{np.median(np.diff(st[:, 0])) * tick_ns:8.0f}")
