#!/usr/bin/env python3
"""Randomised campaign for the fast (default) solver: single problems from 1 to 600 000 correspondences (one workgroup, one
workgroup per 256, grid-stride beyond 4 workgroups per CU) and ragged batches in both batched forms, against the oracle within the
fast mode's tolerances (pose 1e-4, H / b 1e-5 of ref64, equal inlier counts on noise-free data).
usage (GPU box): tools/fuzz_fast_solver.py [seed] [seconds]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
from oracle.oracle import Oracle, Camera as OCam
o32, o64 = Oracle(32), Oracle(64)
ctx = vo.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
big = vo.synth.frame_pair(600000, seed=5)            # one large pair; problems are random subsets of it
jb = np.stack([big["gt_matches"][:, 1], big["model_pairs"][big["gt_matches"][:, 0], 1]], 1).astype(np.int32)
cam_o = OCam(big["rows"], big["cols"], big["z_near"], big["z_far"], big["K"], np.eye(4))
mid = vo.synth.frame_pair(20000, seed=6)             # the batched problems: subsets of this one (every problem gets its own copy of the points)
jm = np.stack([mid["gt_matches"][:, 1], mid["model_pairs"][mid["gt_matches"][:, 0], 1]], 1).astype(np.int32)
cam_m = OCam(mid["rows"], mid["cols"], mid["z_near"], mid["z_far"], mid["K"], np.eye(4))
n_single = n_batch = fails = 0
while time.time() < t_end:
    if rng.integers(0, 3) < 2:
        n = int(rng.choice([rng.integers(16, 300), rng.integers(300, 5000), rng.integers(5000, 70000), rng.integers(70000, 600000)]))
        sel = np.sort(rng.permutation(len(jb))[:n]); j = np.ascontiguousarray(jb[sel]); rounds = int(rng.integers(1, 8))
        keep = bool(rng.integers(0, 2)); thr = float(rng.choice([10000.0, 3.0]))
        s = vo.PICPSolver(ctx); s.setKernelThreshold(thr)
        s.init(vo.Camera(big["rows"], big["cols"], big["z_near"], big["z_far"], big["K"], np.eye(4), ctx=ctx), big["model"], big["cur_pts"])
        s.solve(j, keep, rounds)
        T = s.camera().worldInCameraPose(); H, b = s.system(); n_in = s.numInliers(); s.close()
        r = o64.picp_solve(cam_o, big["model"], big["cur_pts"], j, rounds, thr, keep)
        Hr = r["H"][-1] + np.eye(6); br = r["b"][-1]
        ok = np.abs(T - r["T"]).max() < 1e-4 and abs(n_in - r["num_inliers"]) <= max(2, n // 2000)
        if rounds == 1 and n_in == r["num_inliers"]:      # one round: both sides linearise at the identity, the sums must agree to rounding
            ok = ok and np.abs(H - Hr).max() < 1e-5 * max(1.0, np.abs(Hr).max()) and np.abs(b - br).max() < 1e-5 * max(1.0, np.abs(br).max())
        if not ok:
            print("SINGLE FAIL", n, rounds, keep, thr, float(np.abs(T - r["T"]).max()), n_in, r["num_inliers"]); fails += 1
        n_single += 1
    else:
        P = int(rng.integers(1, 40)); form = int(rng.choice([1, 2])); rounds = int(rng.integers(1, 8)); cap = int(rng.choice([300, 3000, 20000]))
        ns = [int(rng.integers(0, cap + 1)) for _ in range(P)]
        pairs = np.zeros((P, cap, 2), np.int32)
        for p_, n in enumerate(ns):
            pairs[p_, :n] = jm[np.sort(rng.permutation(len(jm))[:n])]
        nw, nm = len(mid["model"]), len(mid["cur_pts"])
        d_pairs = ctx.to_device(pairs); d_n = ctx.to_device(np.array(ns, np.int32))
        d_world = ctx.to_device(np.ascontiguousarray(np.tile(mid["model"], (P, 1)))); d_meas = ctx.to_device(np.ascontiguousarray(np.tile(mid["cur_pts"], (P, 1))))
        d_T = ctx.alloc(P * 64); d_st = ctx.alloc(P * 16)
        K = np.ascontiguousarray(mid["K"].astype(np.float32).T)
        assert ctx.lib.vo_picp_batch_set_form(ctx.h, form) == 0
        rc = ctx.lib.vo_picp_solve_batch_dev(ctx.h, C.c_int(P), C.c_int(mid["rows"]), C.c_int(mid["cols"]), C.c_int(mid["z_near"]), C.c_int(mid["z_far"]),
                                             K.ctypes.data_as(C.c_void_p), C.c_float(10000.0), C.c_int(0), C.c_void_p(d_world), C.c_size_t(nw),
                                             C.c_void_p(d_meas), C.c_size_t(nm), C.c_void_p(d_pairs), C.c_size_t(cap), C.c_void_p(d_n), None,
                                             C.c_int(rounds), C.c_void_p(d_T), C.c_void_p(d_st))
        if rc != 0:
            print("BATCH CALL", rc, ctx.lib.vo_last_error()); fails += 1
        else:
            T = np.zeros((P, 16), np.float32); st = np.zeros((P, 4), np.float32)
            ctx.d2h(T, d_T); ctx.d2h(st, d_st)
            for p_, n in enumerate(ns):
                r = o64.picp_solve(cam_m, mid["model"], mid["cur_pts"], pairs[p_, :n], rounds, 10000.0, False, trace=False)
                Tp = T[p_].reshape(4, 4).T
                tol = 1e-4 if n >= 16 else 1e9
                if np.abs(Tp - r["T"]).max() > tol or (n >= 16 and int(st[p_, 2]) != r["num_inliers"]):
                    print("BATCH FAIL", form, P, cap, n, rounds, float(np.abs(Tp - r["T"]).max()), int(st[p_, 2]), r["num_inliers"]); fails += 1; break
        ctx.lib.vo_picp_batch_set_form(ctx.h, 0)
        for d in (d_pairs, d_n, d_world, d_meas, d_T, d_st): ctx.free(d)
        n_batch += 1
print("single problems", n_single, "batches", n_batch, "failures", fails)
