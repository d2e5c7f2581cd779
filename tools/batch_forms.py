#!/usr/bin/env python3
"""Batched solver: one workgroup per problem (all rounds in one launch) vs one launch per round with the problem as a
grid dimension, over the problem count (vo_picp_batch_set_form)."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import __graft_entry__ as g
vo = g.load_package()
ctx = vo.Context(0)
N = int(os.environ.get("N", "50000")); ITERS = 50
fp = vo.synth.frame_pair(N, seed=2000)
corr = np.stack([fp["gt_matches"][:, 1], fp["gt_matches"][:, 0]], 1).astype(np.int32)
K = np.ascontiguousarray(fp["K"].T.reshape(-1), np.float32)
PS = [int(x) for x in os.environ.get("PS", "1,2,4,8,16,32,48,64,96,128,200").split(",")]
for FORM, P in [(f, p) for f in (1, 2, 0) for p in PS]:
    form = {1: "launch per round ", 2: "one WG per problem", 0: "auto              "}[FORM]
    assert ctx.lib.vo_picp_batch_set_form(ctx.h, FORM) == 0
    d_world = ctx.to_device(np.tile(fp["model"], (P, 1))); d_meas = ctx.to_device(np.tile(fp["cur_pts"], (P, 1)))
    d_pairs = ctx.to_device(np.tile(corr, (P, 1))); d_n = ctx.to_device(np.full(P, N, np.int32))
    d_T = ctx.alloc(P * 64)
    def run():
        rc = ctx.lib.vo_picp_solve_batch_dev(ctx.h, C.c_int(P), C.c_int(480), C.c_int(640), C.c_int(0), C.c_int(10),
                                             K.ctypes.data_as(C.c_void_p), C.c_float(10000.0), C.c_int(0), C.c_void_p(d_world),
                                             C.c_size_t(N), C.c_void_p(d_meas), C.c_size_t(N), C.c_void_p(d_pairs), C.c_size_t(N),
                                             C.c_void_p(d_n), None, C.c_int(ITERS), C.c_void_p(d_T), None)
        assert rc == 0, ctx.lib.vo_last_error()
    for _ in range(2): run()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    T = np.zeros((P, 16), np.float32); ctx.d2h(T, d_T)
    err = float(np.abs(T.reshape(P, 4, 4).transpose(0, 2, 1) - fp["X_gt"]).max())
    print(f"{form} P={P:4d}: {ms:8.3f} ms per call  ({P * ITERS / ms * 1e3:10.0f} iter/s)  err {err:.1e}", flush=True)
    for d in (d_world, d_meas, d_pairs, d_n, d_T): ctx.free(d)
