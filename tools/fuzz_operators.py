#!/usr/bin/env python3
"""Randomised campaign: every operator of the path against the oracle on random sizes (1 .. 9000), drop rates, motions, radii, thresholds;
integer outputs, geometry and the reference-order solver bit for bit.  usage (GPU box): tools/fuzz_operators.py [seed] [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
vo = g.load_package()
from oracle.oracle import Oracle, Camera as OCam
o32 = Oracle(32)
ctx = vo.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 120
it = 0; fails = 0
while time.time() < t_end:
    n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 3000), rng.integers(3000, 9000)]))
    seed = int(rng.integers(0, 1 << 30))
    drop = float(rng.choice([0.0, 0.05, 0.3])); dist = int(rng.integers(0, max(1, n // 8))); md = float(rng.choice([0.0, 0.1, 0.5]))
    ang = float(rng.choice([0.05, 0.3])); tt = float(rng.choice([0.1, 0.6])); noise = float(rng.choice([0.0, 0.5, 2.0]))
    try:
        fp = vo.synth.frame_pair(n, seed=seed, drop=drop if n > 8 else 0.0, distractors=dist, model_drop=md if n > 8 else 0.0, max_angle=ang, max_t=tt, noise_px=noise)
    except Exception as e:
        continue
    radius = float(rng.choice([0.1, 0.03, 0.3]))
    exp_m = o32.match(fp["ref_app"], fp["cur_app"], radius)
    ok = True
    for mode in (1, 2, 3, 4, 5):              # 4 / 5: the exact-duplicate pass first, then 2 / 3 for the queries it leaves open
        ctx.lib.vo_match_set_mode(ctx.h, mode)
        if not np.array_equal(vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], radius, ctx=ctx), exp_m): ok = False; print("MATCH FAIL", n, seed, mode, radius)
    ctx.lib.vo_match_set_mode(ctx.h, 0)
    m = exp_m
    j = vo.extract_correspondences_world(m, fp["model_pairs"], ctx=ctx)
    if not np.array_equal(j, o32.join(m, fp["model_pairs"], linear=True)): ok = False; print("JOIN FAIL", n, seed)
    X = fp["X_gt"]
    xyz, pairs, app = vo.triangulate_points(fp["K"], X, m, fp["ref_pts"], fp["cur_pts"], fp["cur_app"], ctx=ctx)
    e = o32.triangulate(fp["K"], X, m, fp["ref_pts"], fp["cur_pts"], fp["cur_app"])
    if not (np.array_equal(xyz, e[0]) and np.array_equal(pairs, e[1]) and np.array_equal(app, e[2])): ok = False; print("TRI FAIL", n, seed)
    if not np.array_equal(vo.transform_points(X, fp["model"], ctx=ctx), o32.transform_points(X, fp["model"])): ok = False; print("TRANSFORM FAIL", n, seed)
    cam = vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], X, ctx=ctx)
    uv, n_in = cam.projectPoints(fp["model"], keep_indices=bool(it & 1))
    e_uv, e_in = o32.project_points(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], X), fp["model"], keep_indices=bool(it & 1))
    if n_in != e_in or not np.array_equal(uv, e_uv): ok = False; print("PROJECT FAIL", n, seed)
    if len(j):
        thr = float(rng.choice([10000.0, 60.0, 5.0])); keep = bool(rng.integers(0, 2)); rounds = int(rng.integers(1, 12))
        r = o32.picp_solve_raw(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)), fp["model"], fp["cur_pts"], j, rounds, thr, keep)
        s = vo.PICPSolver(ctx); s.setExact(True); s.setKernelThreshold(thr)
        s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
        s.solve(j, keep, rounds)
        T = s.camera().worldInCameraPose(); H, b = s.system()
        if not (np.array_equal(T, r["T"][-1], equal_nan=True) and np.array_equal(H, r["H"][-1], equal_nan=True) and np.array_equal(b, r["b"][-1], equal_nan=True) and s.numInliers() == int(r["stats"][-1, 2])):
            ok = False; print("EXACT SOLVER FAIL", n, seed, thr, keep, rounds)
        s.close()
    it += 1; fails += 0 if ok else 1
print("iterations", it, "failures", fails)
