#!/usr/bin/env python3
"""BASELINE configs[4] end to end, as a distribution instead of one run: the oracle's vo_complete loop
(oracle/vo_pipeline.py, float32 reference-order arithmetic) on the reference's dataset, started from N first relative poses
that differ from the epipolar initialisation's by +-1..4 ulp in every entry of [R | t] -- the size of the difference between
two correct SVD implementations (the reference takes Eigen's JacobiSVD<float>, /root/reference/src/epipolar_utils.cpp:127,133,151).
Prints / writes the distribution of the README metrics (README.md:74-79: 1/r_t 0.47337, RMSE_pos 0.145332, RMSE_points 0.184143).

usage: tools/sweep_first_pose.py [N=400] [seed=1] [out=profiles/r04_first_pose_sweep.json]   (CPU only, ~0.6 s per run)"""
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DATA = os.path.join(ROOT, "tests", "golden", "example_data", "data")
README = dict(median_ratio_inv=0.47337, rmse_position=0.145332, rmse_map=0.184143)


def perturb(X, rng, max_ulp=4):
    Y = X.copy()
    for r in range(3):
        for c in range(4):
            k = int(rng.integers(1, max_ulp + 1)) * (1 if rng.random() < 0.5 else -1)
            v = Y[r, c]
            for _ in range(abs(k)):
                v = np.nextafter(v, np.float32(np.inf if k > 0 else -np.inf))
            Y[r, c] = v
    return Y


def one(args):
    X0, = args
    from oracle import vo_pipeline as vp
    from oracle.oracle import Oracle
    o = Oracle(32)
    res = vp.run_vo_complete(DATA, 100, o, X0=X0)
    ev = vp.evaluate(DATA, res)
    return [ev["median_ratio_inv"], ev["rmse_position"], ev["rmse_map"], ev["mean_orientation_error"], ev["matched"]]


def summarise(a):
    q = np.quantile(a, [0.0, 0.05, 0.25, 0.5, 0.75, 0.95, 1.0])
    return dict(min=float(q[0]), p05=float(q[1]), p25=float(q[2]), median=float(q[3]), p75=float(q[4]), p95=float(q[5]),
                max=float(q[6]), mean=float(np.mean(a)))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r04_first_pose_sweep.json")
    from oracle import vo_pipeline as vp
    from oracle.oracle import Oracle
    base = vp.run_vo_complete(DATA, 100, Oracle(32))
    X = base["trajectory"][1].astype(np.float32)
    ev0 = vp.evaluate(DATA, base)
    rng = np.random.default_rng(seed)
    starts = [perturb(X, rng) for _ in range(n)]
    with Pool(min(7, os.cpu_count() or 1)) as pool:
        rows = np.array(pool.map(one, [(s,) for s in starts], chunksize=4))
    keys = ("median_ratio_inv", "rmse_position", "rmse_map")
    rep = dict(what="oracle vo_complete on example_data from first poses within +-1..4 ulp per entry of the epipolar initialisation's",
               n=n, seed=seed, rounds=100,
               unperturbed={k: ev0[k] for k in keys}, readme=README,
               distribution={k: summarise(rows[:, i]) for i, k in enumerate(keys)},
               readme_inside_central_90={k: bool(np.quantile(rows[:, i], 0.05) <= README[k] <= np.quantile(rows[:, i], 0.95))
                                         for i, k in enumerate(keys)},
               readme_percentile={k: float((rows[:, i] < README[k]).mean()) for i, k in enumerate(keys)},
               samples=[[round(float(v), 6) for v in r[:3]] for r in rows])
    with open(out, "w") as f:
        json.dump(rep, f, indent=1)
    for k in keys:
        d = rep["distribution"][k]
        print(f"{k:18s} README {README[k]:.5f}  unperturbed {ev0[k]:.5f}  p05 {d['p05']:.5f} median {d['median']:.5f} p95 {d['p95']:.5f} "
              f"[{d['min']:.5f}, {d['max']:.5f}]  README at percentile {rep['readme_percentile'][k]:.2f}")


if __name__ == "__main__":
    main()
