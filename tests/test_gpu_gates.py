"""Decisions of the linearisation under rounding: correspondences planted within a fraction of an ulp up to ~1000 ulp of
every gate (camera.h:28 depth, :32-35 image bounds, picp_solver.cpp:78 chi^2 threshold; tests/gate_cases.py), each solved
as its own one-correspondence problem from the SAME pose, so that the statistics of a problem ARE that correspondence's
decision (skipped / inlier / outlier).
  * reference-order arithmetic (form 3) must take the reference's decision on every one of them -- the decision of the
    float32 numpy restatement, whose projection is also checked bit for bit against the oracle;
  * the default arithmetic (forms 1 and 2: one FMA per product, Newton reciprocal) decides on its own, fused values: its
    decision may differ from the reference's only for a correspondence whose reference-order value lies inside a stated
    band around the gate -- BAND below, in ulps of the gate (of the principal point for the image gates at 0): 4 for the
    depth gates, 8 for the image gates, 1024 for chi^2 at thr = 100 (chi^2 = e0^2 + e1^2 inherits the pixel error of (u, v)
    times 2 |e|: 3e-5 relative) -- measured: 2, 3.7, 276.  Outside the band every decision is the reference's.
(A guard that re-evaluates only correspondences near a gate was measured and not kept: its detection arithmetic alone --
margins folded with min3, per-lane slack from |1/z| -- costs the batched solver 9 %, vo_math.h / DESIGN.md section 5.)"""
import ctypes as C

import numpy as np
import pytest

import gate_cases as gc
from oracle.oracle import Camera as OCam

pytestmark = pytest.mark.gpu
BAND = dict(z_far=4, z_near=4, u_lo=8, u_hi=8, v_lo=8, v_hi=8, chi=1024)


def _decisions(ctx, case, form, keep_outliers=0):
    n = len(case["world"])
    lib = ctx.lib
    rows, cols, zn, zf = case["cam"]
    pairs = np.zeros((n, 2), np.int32); npairs = np.ones(n, np.int32)
    T0 = np.tile(np.ascontiguousarray(case["T"].T).ravel(), (n, 1)).astype(np.float32)     # column-major 4x4 per problem
    d = [ctx.to_device(a) for a in (case["world"], case["meas"], pairs, npairs, T0)]
    d_T = ctx.alloc(n * 64); d_stats = ctx.alloc(n * 16)
    K = np.ascontiguousarray(case["K"].T).ravel()
    assert lib.vo_picp_batch_set_form(ctx.h, form) == 0
    try:
        rc = lib.vo_picp_solve_batch_dev(ctx.h, n, rows, cols, zn, zf, K.ctypes.data_as(C.c_void_p), C.c_float(case["thr"]),
                                         keep_outliers, C.c_void_p(d[0]), C.c_size_t(1), C.c_void_p(d[1]), C.c_size_t(1),
                                         C.c_void_p(d[2]), C.c_size_t(1), C.c_void_p(d[3]), C.c_void_p(d[4]), 1,
                                         C.c_void_p(d_T), C.c_void_p(d_stats))
        assert rc == 0, lib.vo_last_error()
        st = np.zeros((n, 4), np.float32)
        ctx.d2h(st, d_stats)
    finally:
        lib.vo_picp_batch_set_form(ctx.h, 0)
        for x in d + [d_T, d_stats]:
            ctx.free(x)
    cls = np.zeros(n, np.int32)
    cls[st[:, 2] == 1.0] = 1                       # one inlier
    cls[(st[:, 2] == 0.0) & (st[:, 1] > 0.0)] = 2  # no inlier, chi of outliers > 0
    return cls, st


@pytest.mark.parametrize("general_k", [False, True])
def test_decisions_at_the_gates(vo, ctx, o32, general_k):
    K = None
    if general_k:                                   # a K with skew and an odd last row entry: the non-pinhole instantiation
        K = np.array([[180.0, 0.7, 320.0], [0.0, 175.0, 240.0], [0.0, 0.0, 1.0]], np.float32)
    case = gc.plant(7000, seed=11 + general_k, K=K)
    n = len(case["world"])
    assert n == 49000
    rows, cols, zn, zf = case["cam"]
    # the numpy restatement IS the oracle's arithmetic: projection bit for bit
    uv, _ = o32.project_points(OCam(rows, cols, zn, zf, case["K"], case["T"]), case["world"], keep_indices=True)
    valid = uv[:, 0] != -1
    assert np.array_equal(valid, case["cls"] != 0)
    assert np.array_equal(uv[valid, 0], case["vals"]["u"][valid]) and np.array_equal(uv[valid, 1], case["vals"]["v"][valid])
    # enough correspondences really sit at the gates, on both sides
    for g, name in enumerate(gc.GATES):
        dg = case["dist"][case["gate"] == g]
        assert (np.abs(dg) <= 1).sum() > 100 and (dg > 16).sum() > 500 and (dg < -16).sum() > 500, name
    exact, _ = _decisions(ctx, case, 3)
    assert np.array_equal(exact, case["cls"]), "reference-order mode: a decision differs from the reference's"
    report = {}
    for form in (2, 1):
        fast, _ = _decisions(ctx, case, form)
        diff = fast != case["cls"]
        for g, name in enumerate(gc.GATES):
            m = diff & (case["gate"] == g)
            worst = float(np.abs(case["dist"][m]).max()) if m.any() else 0.0
            report[(form, name)] = (int(m.sum()), worst)
            assert worst <= BAND[name], f"form {form}, gate {name}: a decision differs {worst} ulp from the gate (band {BAND[name]})"
        # beyond the bands the two arithmetics decide alike on every correspondence (implied by the above; stated for the reader)
        outside = np.abs(case["dist"]) > np.array([BAND[gc.GATES[g]] for g in case["gate"]])
        assert np.array_equal(fast[outside], case["cls"][outside])
    print("default arithmetic: (differing decisions, farthest one in ulp) per (form, gate):", report)
