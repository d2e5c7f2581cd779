"""vo_picp_one_round as the reference calls it (vo_complete.cpp:163-168: a loop of oneRound calls, then camera()):
one launch per call, the finishing launch left to the first getter, the comparison of the pairs overlapped with the
round it guards.  Every arrangement must leave the bits that closed solves on the same arrays leave."""
import numpy as np
import pytest

from oracle.oracle import Camera as OCam

pytestmark = pytest.mark.gpu


def _problem(vo, o32, n=6000, seed=31):
    fp = vo.synth.frame_pair(n, seed=seed, drop=0.0, distractors=0, model_drop=0.0)
    m = o32.match(fp["ref_app"], fp["cur_app"])
    j = np.ascontiguousarray(o32.join(m, fp["model_pairs"]).astype(np.int32))
    assert len(j) > 0.9 * n
    return fp, j


def _solver(vo, ctx, fp, thr=10000.0):
    s = vo.PICPSolver(ctx)
    s.setKernelThreshold(thr)
    s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
    return s


def _state(s):
    H, b = s.system()
    return (s.camera().worldInCameraPose().tobytes(), H.tobytes(), b.tobytes(), s.chiInliers(), s.chiOutliers(), s.numInliers())


def test_open_rounds_are_one_launch_each_and_close_on_the_first_getter(vo, ctx, o32):
    fp, j = _problem(vo, o32)
    s = _solver(vo, ctx, fp)
    for k in range(9):
        s.oneRound(j, False)
        assert s.chainInfo()[0] == k + 1          # open: no finishing launch yet
    open_rounds, spec, redone = s.chainInfo()
    assert (open_rounds, spec, redone) == (9, 8, 0)   # every call after the first went out ahead of its comparison
    got = _state(s)
    assert s.chainInfo()[0] == 0
    t = _solver(vo, ctx, fp)
    t.solve(j, False, 9)
    assert got == _state(t)
    # against the oracle: the usual tolerance of the default arithmetic
    r = o32.picp_solve(OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4)), fp["model"], fp["cur_pts"],
                       j, 9, 10000.0, False, trace=False)
    assert np.abs(np.frombuffer(got[0], np.float32).reshape(4, 4) - r["T"]).max() < 1e-4 and got[5] == r["num_inliers"]
    # getters in the middle of a loop close and reopen: same bits
    u = _solver(vo, ctx, fp)
    for k in range(9):
        u.oneRound(j, False)
        if k in (2, 3, 7):
            u.numInliers()
    assert _state(u) == got
    for x in (s, t, u):
        x.close()


def test_in_place_edit_inside_an_open_chain_repeats_that_round_on_the_new_pairs(vo, ctx, o32):
    fp, j = _problem(vo, o32, seed=32)
    jj = j.copy()
    s = _solver(vo, ctx, fp)
    for _ in range(3):
        s.oneRound(jj, False)
    addr = jj.ctypes.data
    idx = np.arange(1001, 1026)
    jj[idx, 1] = jj[idx + 500, 1]                   # same address, same length, 25 pairs re-pointed at wrong model points
    assert jj.ctypes.data == addr
    for _ in range(3):
        s.oneRound(jj, False)
    assert s.chainInfo() == (6, 5, 1)               # one speculative round found different and repeated
    got = _state(s)
    t = _solver(vo, ctx, fp)
    t.solve(j, False, 3)
    t.solve(jj, False, 3)
    want = _state(t)
    assert got == want
    # and the edit matters: the stale pairs give another answer
    v = _solver(vo, ctx, fp)
    v.solve(j, False, 6)
    assert _state(v)[0] != want[0]
    # an edit of the LAST pair only (the far end of the comparison)
    jj[-1, 1] = jj[0, 1]
    s.oneRound(jj, False)
    t.solve(jj, False, 1)
    assert _state(s) == _state(t)
    for x in (s, t, v):
        x.close()


def test_chain_survives_other_lengths_points_thresholds_and_outlier_policy(vo, ctx, o32):
    fp, j = _problem(vo, o32, seed=33)
    half = np.ascontiguousarray(j[: len(j) // 3])   # another grid: the chain is closed at the old one first
    s, t = _solver(vo, ctx, fp, 50.0), _solver(vo, ctx, fp, 50.0)
    model2 = (fp["model"] + np.float32(0.01)).astype(np.float32)

    def both(pairs, keep, n):
        for _ in range(n):
            s.oneRound(pairs, keep)
        t.solve(pairs, keep, n)

    both(j, False, 2)
    both(half, False, 2)
    both(j, True, 2)                                # keep_outliers flips: parameters re-uploaded between open rounds
    s.setKernelThreshold(20.0); t.setKernelThreshold(20.0)
    both(j, True, 2)
    for x in (s, t):                                # new points under an open chain: re-packed, pose carried on
        x.lib.vo_picp_set_points(x.h, model2.ctypes.data_as(vo.api.C.c_void_p), len(model2),
                                 fp["cur_pts"].ctypes.data_as(vo.api.C.c_void_p), len(fp["cur_pts"]))
    both(j, False, 2)
    assert s.chainInfo()[0] > 0
    assert _state(s) == _state(t)
    # a pose set under an open chain: H, b and the statistics stay those of the last round, the pose is the new one
    both(j, False, 2)
    P = np.eye(4, dtype=np.float32); P[0, 3] = 0.01
    for x in (s, t):
        x.lib.vo_picp_set_pose(x.h, np.ascontiguousarray(P.T).ctypes.data_as(vo.api.C.c_void_p))
    assert _state(s) == _state(t)
    both(j, False, 1)
    assert _state(s) == _state(t)
    s.close(); t.close()


def test_device_side_readers_see_the_rounds_of_an_open_chain(vo, ctx, o32):
    """vo_picp_get_pose_dev / vo_picp_pose_dev_ptr / vo_picp_solve_dev after open rounds: the finishing launch goes first"""
    C = vo.api.C
    fp, j = _problem(vo, o32, seed=34)
    s, t = _solver(vo, ctx, fp), _solver(vo, ctx, fp)
    for _ in range(4):
        s.oneRound(j, False)
    t.solve(j, False, 4)
    want = t.camera().worldInCameraPose()
    d = ctx.alloc(64)
    assert s.lib.vo_picp_get_pose_dev(s.h, C.c_void_p(d)) == 0
    got = np.zeros(16, np.float32); ctx.d2h(got, d)
    assert np.array_equal(got.reshape(4, 4).T, want)
    for _ in range(2):
        s.oneRound(j, False)
    t.solve(j, False, 2)
    p = C.c_void_p()
    assert s.lib.vo_picp_pose_dev_ptr(s.h, C.byref(p)) == 0 and s.chainInfo()[0] == 0
    ctx.synchronize()
    ctx.d2h(got, p.value)
    assert np.array_equal(got.reshape(4, 4).T, t.camera().worldInCameraPose())
    # device pairs after open rounds
    dj = ctx.to_device(j)
    for x in (s, t):
        x.oneRound(j, False)
        assert x.lib.vo_picp_solve_dev(x.h, C.c_void_p(dj), len(j), None, 0, 3) == 0
    assert _state(s) == _state(t)
    # exact mode switched under an open chain: the open rounds are closed in the arithmetic they ran in
    s.oneRound(j, False); t.oneRound(j, False); t.numInliers()
    s.setExact(True); t.setExact(True)
    s.oneRound(j, False); t.oneRound(j, False)
    assert _state(s) == _state(t)
    ctx.free(d); ctx.free(dj)
    s.close(); t.close()


def test_small_and_empty_problems_keep_their_one_launch_forms(vo, ctx, o32):
    fp, j = _problem(vo, o32, n=200, seed=35)
    s, t = _solver(vo, ctx, fp), _solver(vo, ctx, fp)
    for _ in range(5):
        s.oneRound(j, False)
    assert s.chainInfo()[0] == 0                    # one workgroup: every call is complete by itself
    t.solve(j, False, 5)
    assert _state(s) == _state(t)
    e = np.zeros((0, 2), np.int32)
    for _ in range(3):
        s.oneRound(e, False)
    t.solve(e, False, 3)
    assert _state(s) == _state(t)
    s.close(); t.close()


def test_rounds_enqueued_ahead_of_the_caller_are_claimed_repeated_or_ignored(vo, ctx, o32):
    """Once two calls in a row have matched, a call enqueues its round and up to seven more as one graph launch (capi.hip:
    run_ahead; the graph is captured the third time its launch geometry asks for it); the following calls claim theirs after
    comparing.  Whatever interrupts the loop at whatever position of that window -- a getter, an in-place edit, other points,
    another threshold, a closed solve -- must leave the bits of closed solves on the same arrays: rounds that ran ahead
    unclaimed are repeated on the new data or ignored.  ONE pair of handles goes through all of it (the graphs, the remembered
    loop length and the streak are the handle's), the twin only ever through closed solves."""
    fp, j = _problem(vo, o32, n=5000, seed=36)
    model = [fp["model"], (fp["model"] + np.float32(0.02)).astype(np.float32)]
    C = vo.api.C
    s, t = _solver(vo, ctx, fp, 200.0), _solver(vo, ctx, fp, 200.0)
    for _ in range(5):                              # the loop shows itself, its windows get their graphs
        for _ in range(20):
            s.oneRound(j, False)
        t.solve(j, False, 20)
        assert _state(s) == _state(t)
    assert s.graphInfo()[1] >= 2 and s.graphInfo()[2] == 0          # windows are graph launches now (vo_picp_graph_info)
    which, thr = 0, 200.0
    for k in range(2, 21):                          # the interruption after k calls: every position of a window, twice over
        for what in ("getter", "edit", "points", "threshold", "solve"):
            jj = j.copy()
            for _ in range(k):
                s.oneRound(jj, False)
            t.solve(j, False, k)
            if what == "getter":
                assert s.numInliers() == t.numInliers()
            elif what == "edit":
                idx = np.arange(700, 720)
                jj[idx, 1] = jj[idx + 900, 1]
            elif what == "points":
                which ^= 1
                for x in (s, t):
                    assert x.lib.vo_picp_set_points(x.h, model[which].ctypes.data_as(C.c_void_p), len(model[which]),
                                                    fp["cur_pts"].ctypes.data_as(C.c_void_p), len(fp["cur_pts"])) == 0
            elif what == "threshold":
                thr = 20.0 if thr == 200.0 else 200.0
                s.setKernelThreshold(thr); t.setKernelThreshold(thr)
            else:
                s.solve(jj, False, 3); t.solve(jj, False, 3)
            for _ in range(9):
                s.oneRound(jj, False)
            t.solve(jj, False, 9)
            assert _state(s) == _state(t), (k, what)
    assert s.chainInfo()[2] >= 19                   # the edits were found by the comparison and their rounds repeated
    s.close(); t.close()
    # VO_PICP_RUN_AHEAD=0 in the environment is read when a handle is made: one round per call, same bits
    import os
    os.environ["VO_PICP_RUN_AHEAD"] = "0"
    try:
        u = _solver(vo, ctx, fp, 200.0)
    finally:
        os.environ.pop("VO_PICP_RUN_AHEAD", None)
    w = _solver(vo, ctx, fp, 200.0)
    for _ in range(4):
        for _ in range(19):
            u.oneRound(j, False); w.oneRound(j, False)
        assert _state(u) == _state(w)
    assert u.graphInfo()[1] == 0 and w.graphInfo()[1] >= 1
    u.close(); w.close()
