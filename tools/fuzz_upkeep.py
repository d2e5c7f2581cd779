#!/usr/bin/env python3
"""Randomised campaign for round 5's pieces.   usage (GPU box): tools/fuzz_upkeep.py [seed] [seconds]
  maps        random runs of map.update (vo_map_*): clouds of 0..6000 rows drawn from a pool (so that they overlap the map and
              themselves), components zeroed with either sign, rows with NaN, optional isometries, maps that start too small
              -- against the oracle's first-occurrence dictionary, and for small runs against the reference's double loop
              as written (oracle/vo_pipeline.py: Map, literal_update): entries, order, point and appearance BITS.
  chains      random walks over a PICPSolver handle -- oneRound on the same / an edited-in-place / another array, getters in
              between or not, solve(n), new points, threshold and outlier policy, pose resets -- against a twin handle that
              closes every round by itself (one round, then a getter): pose, H, b, statistics bit for bit, both arithmetics.
  sequences   random short sequences with the map kept inside the device-resident chain (SequencePipeline(keep_map), solver in
              reference-order arithmetic) against the oracle's loop: trajectory, map entries and history bit for bit.
  inits       random first pairs: vo_estimate_transform_dev (sums on the GPU) against the oracle's numpy restatement."""
import os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
from oracle.oracle import Oracle, Camera as OCam
from oracle import vo_pipeline as vp
o32 = Oracle(32)
ctx = vo.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
n_maps = n_chains = n_seqs = n_inits = fails = 0


def same_map(m, pts, app):
    p, a = m.read()
    return (len(p) == len(pts) and p.tobytes() == np.array(pts, np.float32).reshape(-1, 3).tobytes()
            and a.tobytes() == np.array(app, np.float32).reshape(-1, 10).tobytes())


def fuzz_map():
    pool_n = int(rng.choice([30, 400, 5000]))
    pool = (np.round(rng.uniform(-1, 1, (pool_n, 10)), int(rng.choice([1, 3]))) if rng.integers(0, 2) else rng.uniform(-1, 1, (pool_n, 10))).astype(np.float32)
    m = vo.Map(ctx, capacity=int(rng.choice([0, 64, 20000])))
    dic = vp.Map()
    lit_p, lit_a = [], []
    literal = pool_n <= 400
    ok = True
    for _ in range(int(rng.integers(1, 7))):
        n = int(rng.choice([0, 1, 2, 63, 64, 65, 255, 256, 257, 1000, 6000])) if not literal else int(rng.choice([0, 1, 2, 63, 65, 257, 600]))
        idx = rng.integers(0, pool_n, n)
        a = pool[idx].copy()
        if n:
            z = rng.random(a.shape) < float(rng.choice([0.0, 0.05, 0.5]))
            a[z] = np.where(rng.random(int(z.sum())) < 0.5, np.float32(0.0), np.float32(-0.0))
            bad = rng.random(n) < float(rng.choice([0.0, 0.03]))
            a[bad, rng.integers(0, 10, int(bad.sum()))] = np.nan
        p = rng.normal(0, 3, (n, 3)).astype(np.float32)
        T = vo.synth.random_isometry(rng, 1.0, 2.0).astype(np.float32) if rng.integers(0, 2) else None
        m.update(p, a, T)
        moved = o32.transform_points(T, p) if (T is not None and n) else p
        dic.update(list(moved), list(a))
        if literal:
            vp.literal_update(lit_p, lit_a, list(moved), list(a))
        ok = ok and same_map(m, dic.pts, dic.app) and (not literal or same_map(m, lit_p, lit_a))
    m.close()
    return ok


def state(s):
    H, b = s.system()
    return (s.camera().worldInCameraPose().tobytes(), H.tobytes(), b.tobytes(), s.chiInliers(), s.chiOutliers(), s.numInliers())


def fuzz_chain():
    n = int(rng.choice([90, 300, 2000, 7000]))
    fp = vo.synth.frame_pair(n, seed=int(rng.integers(0, 1 << 30)), distractors=n // 20, model_drop=0.1)
    m = o32.match(fp["ref_app"], fp["cur_app"])
    j = np.ascontiguousarray(o32.join(m, fp["model_pairs"]).astype(np.int32))
    if len(j) < 20:
        return True
    exact = bool(rng.integers(0, 2))
    thr = float(rng.choice([10000.0, 50.0, 5.0]))
    cam = vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx)
    s, t = vo.PICPSolver(ctx), vo.PICPSolver(ctx)
    for x in (s, t):
        x.setExact(exact); x.setKernelThreshold(thr); x.init(cam, fp["model"], fp["cur_pts"])
    cur = j.copy()
    keep = False
    ok = True
    if rng.integers(0, 2):                            # the loop shows itself first: windows of rounds ahead of the caller get their
        for _ in range(4):                            # graphs (captured the third time a geometry asks), the loop length is remembered
            k = int(rng.integers(9, 20))
            for _ in range(k):
                s.oneRound(cur, keep)
            t.solve(cur, keep, k)
            ok = ok and state(s) == state(t)
    for _ in range(int(rng.integers(3, 40))):
        op = int(rng.integers(0, 14))
        if op >= 10:
            op = 0                                    # (half of the steps are the loop's own call)
        if op <= 3:                                   # the loop's call, on whatever the array holds now
            s.oneRound(cur, keep)
            t.solve(cur, keep, 1); t.numInliers()      # the twin closes every round
        elif op == 4:                                 # in-place edit: a few pairs re-pointed
            k = rng.integers(0, len(cur), 5)
            cur[k, 1] = cur[rng.integers(0, len(cur), 5), 1]
        elif op == 5:                                 # another array (another length: another grid, maybe the one-workgroup form)
            cur = np.ascontiguousarray(j[: int(rng.integers(1, len(j) + 1))].copy())
        elif op == 6:
            ok = ok and state(s) == state(t)
        elif op == 7:
            k = int(rng.integers(2, 6))
            s.solve(cur, keep, k); t.solve(cur, keep, k)
        elif op == 8:
            keep = not keep
            thr = float(rng.choice([10000.0, 50.0, 5.0]))
            s.setKernelThreshold(thr); t.setKernelThreshold(thr)
        else:
            if rng.integers(0, 2):
                for x in (s, t):
                    x.init(cam, fp["model"], fp["cur_pts"])
            else:
                P = np.ascontiguousarray(vo.synth.random_isometry(rng, 0.02, 0.05).astype(np.float32).T)
                for x in (s, t):
                    assert x.lib.vo_picp_set_pose(x.h, P.ctypes.data_as(C.c_void_p)) == 0
    ok = ok and state(s) == state(t)
    s.close(); t.close()
    return ok


def fuzz_sequence():
    F = int(rng.integers(3, 9)); nv = int(rng.choice([60, 200, 700])); rounds = int(rng.integers(2, 20)); seed = int(rng.integers(0, 1 << 30))
    seq = vo.synth.sequence(seed=seed, n_frames=F, n_visible=nv)
    fr = seq["frames"]
    if min(len(fr[0]["pts"]), len(fr[1]["pts"])) < 12:
        return True
    sp = vo.SequencePipeline(ctx, seq, n_iters=rounds, exact=True, keep_map=True, map_capacity=int(rng.choice([16, 100000])))
    sp.run()
    traj = sp.trajectory()
    pts, app = sp.map.read()
    hist = sp.map.history()
    sp.close()
    res = vp.run_sequence([(f["pts"], f["app"]) for f in fr], seq["K"], seq["rows"], seq["cols"], seq["z_near"], seq["z_far"], rounds, o32, X0=traj[1])
    ref = np.array(res["trajectory"], np.float32)
    h = vp.iso_inv32(traj[1])
    for X in traj[2:]:
        h = vp.iso_mul32(h, vp.iso_inv32(X))
    mm = res["map"]
    return (np.array_equal(traj, ref, equal_nan=True) and len(pts) == len(mm.pts) and pts.tobytes() == np.array(mm.pts, np.float32).reshape(-1, 3).tobytes()
            and app.tobytes() == np.array(mm.app, np.float32).reshape(-1, 10).tobytes() and np.array_equal(hist, h, equal_nan=True))


def fuzz_init():
    nv = int(rng.choice([40, 300, 3000])); seed = int(rng.integers(0, 1 << 30))
    seq = vo.synth.sequence(seed=seed, n_frames=2, n_visible=nv, noise_px=float(rng.choice([0.0, 0.2])))
    f0, f1 = seq["frames"]
    corr = o32.match(f0["app"], f1["app"])
    if len(corr) < 12:
        return True
    d_c, d_a, d_b = ctx.to_device(corr), ctx.to_device(f0["pts"]), ctx.to_device(f1["pts"])
    d_n = ctx.to_device(np.array([len(corr)], np.int32))
    X = np.zeros(16, np.float32)
    K = np.ascontiguousarray(np.asarray(seq["K"], np.float32).T).ravel()
    rc = ctx.lib.vo_estimate_transform_dev(ctx.h, K.ctypes.data_as(C.c_void_p), C.c_void_p(d_c), C.c_int(len(corr) + 5 if False else len(corr)), C.c_void_p(d_n),
                                           C.c_void_p(d_a), C.c_int(len(f0["pts"])), C.c_void_p(d_b), C.c_int(len(f1["pts"])), X.ctypes.data_as(C.c_void_p))
    for d in (d_c, d_a, d_b, d_n):
        ctx.free(d)
    if rc != 0:
        print("INIT ERROR", ctx.lib.vo_last_error())
        return False
    Xo = vp.estimate_transform(o32, seq["K"], corr, f0["pts"], f1["pts"])
    X = X.reshape(4, 4).T
    # both solve the same 9 x 9 null-space problem in double and round once; noisy pairs condition it less well
    return bool(np.abs(X - Xo).max() < 5e-4 * max(1.0, float(np.abs(Xo[:3, 3]).max())))


while time.time() < t_end:
    kind = int(rng.integers(0, 8))
    try:
        if kind <= 2:
            ok = fuzz_map(); n_maps += 1; what = "MAP"
        elif kind <= 5:
            ok = fuzz_chain(); n_chains += 1; what = "CHAIN"
        elif kind == 6:
            ok = fuzz_sequence(); n_seqs += 1; what = "SEQUENCE"
        else:
            ok = fuzz_init(); n_inits += 1; what = "INIT"
    except vo.VoError as e:
        print("ERROR", kind, e); fails += 1; continue
    if not ok:
        print(what, "FAIL after", n_maps, n_chains, n_seqs, n_inits); fails += 1
print("maps", n_maps, "chains", n_chains, "sequences", n_seqs, "inits", n_inits, "failures", fails)
