#!/usr/bin/env python3
"""vo_picp_solve_batch_dev, 200 problems x 50k x 50 rounds, one workgroup per problem: ms per call, no checks (for builds
whose results are wrong on purpose: VO_HIP_LIB=... experiments)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import __graft_entry__ as g
vo = g.load_package()
ctx = vo.Context(0)
N = int(os.environ.get("N", "50000")); ITERS = 50; P = int(os.environ.get("P", "200"))
fp = vo.synth.frame_pair(N, seed=2000)
corr = np.stack([fp["gt_matches"][:, 1], fp["gt_matches"][:, 0]], 1).astype(np.int32)
K = np.ascontiguousarray(fp["K"].T.reshape(-1), np.float32)
assert ctx.lib.vo_picp_batch_set_form(ctx.h, 2) == 0
d_world = ctx.to_device(np.tile(fp["model"], (P, 1))); d_meas = ctx.to_device(np.tile(fp["cur_pts"], (P, 1)))
d_pairs = ctx.to_device(np.tile(corr, (P, 1))); d_n = ctx.to_device(np.full(P, N, np.int32))
d_T = ctx.alloc(P * 64)
def run():
    rc = ctx.lib.vo_picp_solve_batch_dev(ctx.h, C.c_int(P), C.c_int(480), C.c_int(640), C.c_int(0), C.c_int(10),
                                         K.ctypes.data_as(C.c_void_p), C.c_float(10000.0), C.c_int(0), C.c_void_p(d_world),
                                         C.c_size_t(N), C.c_void_p(d_meas), C.c_size_t(N), C.c_void_p(d_pairs), C.c_size_t(N),
                                         C.c_void_p(d_n), None, C.c_int(ITERS), C.c_void_p(d_T), None)
    assert rc == 0
for _ in range(3): run()
ctx.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10): run()
    ctx.synchronize()
    best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
print(f"{os.path.basename(os.environ.get('VO_HIP_LIB', 'libvo_hip.so'))}: {best:.3f} ms per call (gather + {ITERS} rounds, P={P})", flush=True)
