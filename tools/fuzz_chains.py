#!/usr/bin/env python3
"""Randomised campaign for the device-resident chains: random short sequences (SequencePipeline, solver in reference-order
arithmetic) against the oracle's loop -- counts, finite poses and NaN positions bit for bit -- and random batches of independent
frames (vo_frames_batch_dev, solver form 3) against the oracle frame by frame.  usage (GPU box): tools/fuzz_chains.py [seed] [seconds]"""
import os, sys, time
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
n_seq = n_batch = fails = 0
while time.time() < t_end:
    if rng.integers(0, 2) == 0:
        F = int(rng.integers(3, 9)); nv = int(rng.choice([60, 200, 700, 1500])); rounds = int(rng.integers(2, 25)); seed = int(rng.integers(0, 1 << 30))
        seq = vo.synth.sequence(seed=seed, n_frames=F, n_visible=nv)
        fr = seq["frames"]
        kind = int(rng.integers(0, 4))
        if kind == 1 and F > 4:
            t = int(rng.integers(2, F)); fr[t] = dict(ids=np.zeros(0, np.int64), pts=np.zeros((0, 2), np.float32), app=np.zeros((0, 10), np.float32))
        if kind == 2 and F > 4:
            t = int(rng.integers(2, F)); k = int(rng.integers(1, 6)); fr[t] = dict(ids=fr[t]["ids"][:k], pts=fr[t]["pts"][:k].copy(), app=fr[t]["app"][:k].copy())
        if min(len(fr[0]["pts"]), len(fr[1]["pts"])) < 12:
            continue
        try:
            sp = vo.SequencePipeline(ctx, seq, n_iters=rounds, exact=True)
            sp.run()
            traj, counts = sp.trajectory(), sp.counts()
            sp.close()
        except vo.VoError as e:
            print("SEQ ERROR", seed, F, nv, kind, e); fails += 1; continue
        res = vp.run_sequence([(f["pts"], f["app"]) for f in fr], seq["K"], seq["rows"], seq["cols"], seq["z_near"], seq["z_far"], rounds, o32, X0=traj[1], keep_map=False)
        ref = np.array(res["trajectory"], np.float32)
        ok = (np.array_equal(counts[2:, :2], np.array([s[:2] for s in res["stats"]]).reshape(-1, 2)) and np.array_equal(counts[1:, 2], np.array(res["tri_counts"]))
              and np.array_equal(np.isnan(traj), np.isnan(ref)) and np.array_equal(traj, ref, equal_nan=True))
        if not ok:
            print("SEQ FAIL", seed, F, nv, kind, rounds); fails += 1
        n_seq += 1
    else:
        F = int(rng.integers(1, 20)); n = int(rng.choice([100, 256, 257, 900, 3000])); rounds = int(rng.integers(1, 15)); seed = int(rng.integers(0, 1 << 20)) * 3
        fps = []
        for i in range(F):
            f = vo.synth.frame_pair(n, seed=seed + 33 * i, distractors=int(n // 20), noise_px=float(rng.choice([0.0, 0.5])))
            keep = np.sort(rng.permutation(n)[: n - n // 7]); f["model_pairs"] = np.ascontiguousarray(f["model_pairs"][keep]); fps.append(f)
        if len({(len(f["ref_app"]), len(f["cur_app"])) for f in fps}) != 1:
            continue
        assert ctx.lib.vo_picp_batch_set_form(ctx.h, 3) == 0
        bp = vo.BatchPipeline(ctx, fps, n_iters=rounds)
        bp.run()
        T, st, c = bp.poses(), bp.stats(), bp.counts()
        ok = True
        for i, f in enumerate(fps):
            m = o32.match(f["ref_app"], f["cur_app"]); j = o32.join(m, f["model_pairs"])
            if not (np.array_equal(bp.fetch("match", i), m) and np.array_equal(bp.fetch("join", i), j)): ok = False; break
            r = o32.picp_solve(OCam(480, 640, 0, 10, f["K"], np.eye(4)), f["model"], f["cur_pts"], j, rounds, 10000.0, False, trace=False)
            if not (np.array_equal(T[i], r["T"].astype(np.float32)) and int(st[i, 2]) == r["num_inliers"]): ok = False; break
            x, p_, a_ = o32.triangulate(f["K"], T[i], m, f["ref_pts"], f["cur_pts"], f["cur_app"])
            if not (np.array_equal(bp.fetch("tri_pairs", i), p_) and np.array_equal(bp.fetch("tri_xyz", i), x) and np.array_equal(bp.fetch("tri_app", i), a_)): ok = False; break
        bp.close()
        ctx.lib.vo_picp_batch_set_form(ctx.h, 0)
        if not ok:
            print("BATCH FAIL", seed, F, n, rounds); fails += 1
        n_batch += 1
print("sequences", n_seq, "batches", n_batch, "failures", fails)
