#!/usr/bin/env python3
"""Rounds per second of the solver in reference-order arithmetic (vo_picp_set_exact) on one 50k-correspondence pair and, batched,
on 256 pairs (one workgroup per problem); the result is compared bit for bit with the CPU restatement.  usage: tools/exact_rate.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
from oracle.oracle import Oracle, Camera as OCam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
fp = vo.synth.frame_pair(n, seed=2000)
ctx = vo.Context(0)
j = vo.extract_correspondences_world(fp["gt_matches"], fp["model_pairs"], ctx=ctx)
s = vo.PICPSolver(ctx)
s.setExact(True); s.setKernelThreshold(10000.0)
cam = vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx)
s.init(cam, fp["model"], fp["cur_pts"])
s.setCorrespondences(j)
for rounds in (10, 50):
    s.init(cam, fp["model"], fp["cur_pts"]); s.setCorrespondences(j)
    s.rounds(False, 2); s.camera()
    s.init(cam, fp["model"], fp["cur_pts"]); s.setCorrespondences(j)
    t0 = time.perf_counter(); s.rounds(False, rounds); T = s.camera().worldInCameraPose(); dt = time.perf_counter() - t0
    print(f"exact, {len(j)} correspondences: {rounds} rounds in {dt*1e3:.2f} ms = {dt/rounds*1e6:.1f} us per round = {rounds/dt:.0f} iter/s", flush=True)
r = Oracle(32).picp_solve_raw(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)), fp["model"], fp["cur_pts"], j, 50, 10000.0, False)
print("bit-identical to the CPU restatement after 50 rounds:", np.array_equal(T, r["T"][-1]), "pose err vs gt", float(np.abs(T - fp["X_gt"]).max()))
