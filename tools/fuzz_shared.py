#!/usr/bin/env python3
"""Randomised campaign for the batched solver's shared form (csrc/picp.hip, picp_batch_shared_kernel: the waves of the CUs
without a problem take chunks of the problems' correspondences every round).   usage (GPU box): tools/fuzz_shared.py [seed] [seconds]
Random problem counts (1 .. 0.65 per CU), capacities (18 432 .. 40 000), per-problem sizes from empty to full, starting poses,
thresholds, outlier policy, projection (pinhole / with skew), round counts; per case
  * the call twice: the same bits (helper waves arrive when they arrive);
  * the call with every helper wave absent (VO_PICP_HELP_ABSENT: the problems' own workgroups stand in for all chunks): the
    same bits;
  * problems that share data and starting pose: the same bits;
  * against the call in its other forms (one launch per round; one workgroup per problem without helpers; other chunks):
    poses within 3e-5 and the same inliers when no threshold is in play, within 3e-3 and a few flipped inliers otherwise;
  * two problems per case against the oracle: poses within 1e-4 (likewise)."""
import os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
from oracle.oracle import Oracle, Camera as OCam
o32 = Oracle(32)
ctx = vo.Context(0)
lib = ctx.lib
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
for k in ("VO_PICP_SHARE", "VO_PICP_HELP_KEEP", "VO_PICP_HELP_G", "VO_PICP_HELP_SLACK", "VO_PICP_HELP_ABSENT"):
    os.environ.pop(k, None)
cases = fails = 0
pool = {}


def frame(n):
    if n not in pool:
        fp = vo.synth.frame_pair(n, seed=9000 + n, distractors=n // 40)
        mp = dict(fp["model_pairs"].tolist())
        fp["corr"] = np.array([(c, mp[r]) for r, c in fp["gt_matches"].tolist()], np.int32)
        pool[n] = fp
    return pool[n]


while time.time() < t_end:
    n = int(rng.choice([18432, 19000, 24577, 30000, 40000]))
    fp = frame(n)
    P = int(rng.choice([1, 2, 5, 13, 40, 97, 166, int(rng.integers(1, 167))]))
    iters = int(rng.choice([1, 2, 3, 7, 20]))
    thr = float(rng.choice([10000.0, 60.0, 8.0]))
    keep = int(rng.integers(0, 2))
    kind = int(rng.integers(0, 4))
    if kind == 0: sizes = np.full(P, n)
    elif kind == 1: sizes = rng.integers(0, n + 1, P)
    elif kind == 2: sizes = rng.choice([0, 1, 5, 6143, 6144, 6145, 12288, n - 1, n], P)
    else: sizes = np.where(rng.random(P) < 0.8, rng.integers(0, 3000, P), n)      # a few long problems among short ones
    sizes = sizes.astype(np.int32)
    T0 = np.stack([vo.synth.random_isometry(rng, 0.01, 0.02) for _ in range(P)]).astype(np.float32)
    twin = None
    if P >= 2:                                               # two problems with the same data
        a, b = rng.choice(P, 2, replace=False)
        sizes[b] = sizes[a]; T0[b] = T0[a]; twin = (int(a), int(b))
    Km = np.asarray(fp["K"], np.float32).copy()
    if rng.integers(0, 3) == 0:
        Km[0, 1] = 0.5
    K = np.ascontiguousarray(Km.T).ravel()
    d = [ctx.to_device(np.tile(fp["model"], (P, 1))), ctx.to_device(np.tile(fp["cur_pts"], (P, 1))), ctx.to_device(np.tile(fp["corr"], (P, 1))),
         ctx.to_device(sizes), ctx.to_device(np.ascontiguousarray(np.transpose(T0, (0, 2, 1))).reshape(P, 16))]
    d_T, d_S = ctx.alloc(P * 64), ctx.alloc(P * 16)

    def run(form=2, **env):
        for k, v in env.items():
            os.environ[k] = str(v)
        assert lib.vo_picp_batch_set_form(ctx.h, form) == 0
        rc = lib.vo_picp_solve_batch_dev(ctx.h, P, 480, 640, 0, 10, K.ctypes.data_as(C.c_void_p), C.c_float(thr), keep, C.c_void_p(d[0]),
                                         C.c_size_t(len(fp["model"])), C.c_void_p(d[1]), C.c_size_t(len(fp["cur_pts"])), C.c_void_p(d[2]),
                                         C.c_size_t(n), C.c_void_p(d[3]), C.c_void_p(d[4]), iters, C.c_void_p(d_T), C.c_void_p(d_S))
        for k in env:
            os.environ.pop(k)
        lib.vo_picp_batch_set_form(ctx.h, 0)
        assert rc == 0, lib.vo_last_error()
        f = C.c_int(); lib.vo_picp_batch_info(ctx.h, C.byref(f), None)
        T = np.zeros((P, 16), np.float32); S = np.zeros((P, 4), np.float32)
        ctx.d2h(T, d_T); ctx.d2h(S, d_S)
        return T, S, f.value

    ref = run()
    why = []
    if ref[2] != 4: why.append("form %d" % ref[2])
    again = run()
    if again[0].tobytes() != ref[0].tobytes() or again[1].tobytes() != ref[1].tobytes(): why.append("run to run")
    alone = run(VO_PICP_HELP_ABSENT=1)
    if alone[0].tobytes() != ref[0].tobytes() or alone[1].tobytes() != ref[1].tobytes(): why.append("helpers absent")
    if twin and (ref[0][twin[0]].tobytes() != ref[0][twin[1]].tobytes() or ref[1][twin[0]].tobytes() != ref[1][twin[1]].tobytes()): why.append("twins")
    others = [run(form=1), run(VO_PICP_SHARE=0), run(VO_PICP_HELP_KEEP=int(rng.integers(2, 9)), VO_PICP_HELP_G=int(rng.integers(1, 9)))]
    # Another summation order gives rounding-level differences -- unless correspondences sit AT the kernel threshold (tight
    # thresholds, first rounds from a perturbed pose: thousands do) and flip between inlier and outlier, which changes the
    # system by their whole terms.  Without a threshold in play: 3e-5 and the same inliers; with one: 3e-3 and a few flips
    # (a chunk lost or taken twice would move thousands of inliers and the pose by 1e-2).
    # Where most correspondences are OUTLIERS (threshold 8 against the ~6 px the perturbed starting pose is off: a few dozen
    # inliers hold the six unknowns, and two flips move the pose by 1e-2) the forms are not compared at all: that regime is
    # covered by the exact properties above.
    def close(dT, dn, size, tight, inliers):
        if thr >= 10000.0: return dn == 0 and dT < tight
        if inliers < size // 2: return True
        return dn <= max(3, size // 60) and dT < 3e-3      # (0.6 % flips seen at threshold 8 after two rounds from a perturbed pose)
    for name, x in zip(("rounds", "plain", "chunks"), others):
        for p in range(P):
            dT, dn = float(np.abs(x[0][p] - ref[0][p]).max()), abs(float(x[1][p, 2] - ref[1][p, 2]))
            if not close(dT, dn, sizes[p], 3e-5, int(ref[1][p, 2])):
                why.append("%s p%d n%d dT %.1e dn %d" % (name, p, sizes[p], dT, dn)); break
    big = [p for p in range(P) if sizes[p] >= 50]            # (a handful of correspondences: 6 unknowns held by the damping alone)
    for p in rng.choice(big, min(len(big), 2), replace=False) if big else []:
        r = o32.picp_solve(OCam(480, 640, 0, 10, Km, T0[p]), fp["model"], fp["cur_pts"], fp["corr"][: sizes[p]], iters, thr, bool(keep), trace=False)
        dT, dn = float(np.abs(ref[0][p].reshape(4, 4).T - r["T"]).max()), abs(int(ref[1][p, 2]) - r["num_inliers"])
        if not close(dT, dn, sizes[p], 1e-4, int(ref[1][p, 2])): why.append("oracle p%d n%d dT %.1e dn %d" % (p, sizes[p], dT, dn))
    for x in d + [d_T, d_S]:
        ctx.free(x)
    cases += 1
    if why:
        fails += 1
        print("FAIL", dict(n=n, P=P, iters=iters, thr=thr, keep=keep, kind=kind), why, flush=True)
print("cases", cases, "failures", fails)
