#!/usr/bin/env python3
"""Round time of the solver on a problem of the reference dataset's size (<= 256 correspondences): all rounds in one launch
(default) against the launch-per-round form (VO_PICP_SMALL=0).  usage: [VO_PICP_SMALL=0] tools/small_rate.py [n] [rounds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 127
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
fp = vo.synth.frame_pair(max(n, 8), seed=9700)
corr = np.stack([fp["gt_matches"][:n, 1], fp["model_pairs"][fp["gt_matches"][:n, 0], 1]], 1).astype(np.int32)
ctx = vo.Context(0)
s = vo.PICPSolver(ctx)
s.setKernelThreshold(10000.0)
exact = os.environ.get("EXACT", "0") == "1"
s.setExact(exact)
cam = vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx)
for rep in range(3):
    s.init(cam, fp["model"], fp["cur_pts"]); s.setCorrespondences(corr)
    t0 = time.perf_counter(); s.rounds(False, rounds); T = s.camera().worldInCameraPose(); dt = time.perf_counter() - t0
print(f"exact={int(exact)} VO_PICP_SMALL={os.environ.get('VO_PICP_SMALL', '1')}: {n} correspondences, {rounds} rounds: {dt*1e3:.2f} ms = {dt/rounds*1e6:.2f} us per round; pose err {np.abs(T - fp['X_gt']).max():.1e}")
