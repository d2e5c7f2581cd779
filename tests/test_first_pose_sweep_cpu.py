"""BASELINE configs[4] end to end (README.md:74-79 of the reference: 1/r_t 0.47337, RMSE_pos 0.145332, RMSE_points 0.184143) as a
DISTRIBUTION: the chain of 119 PICP solves with 6..32 inliers each amplifies a last-bit change of the first relative pose, so one run
is an anecdote.  tools/sweep_first_pose.py runs the oracle's vo_complete loop from 1000 first poses within +-1..4 ulp per entry of the
epipolar initialisation's (the difference between two correct SVDs; the reference uses Eigen's JacobiSVD<float>,
epipolar_utils.cpp:127,133,151) and commits the distribution; here: the README values lie inside its central 90 %, and the committed
samples are what the script produces today."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_readme_metrics_lie_inside_the_sweep():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_first_pose_sweep.json")))
    assert d["n"] >= 200
    s = np.array(d["samples"])
    for i, k in enumerate(("median_ratio_inv", "rmse_position", "rmse_map")):
        lo, hi = np.quantile(s[:, i], [0.05, 0.95])
        assert lo <= d["readme"][k] <= hi, (k, lo, d["readme"][k], hi)
        assert d["readme_inside_central_90"][k]
    # the spread itself: the map error of this chain is not a number, it is a range
    assert np.quantile(s[:, 2], 0.95) > 2 * np.quantile(s[:, 2], 0.05)


def test_committed_samples_are_reproducible(o32):
    import sweep_first_pose as sw
    from oracle import vo_pipeline as vp
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_first_pose_sweep.json")))
    X = vp.run_vo_complete(sw.DATA, 100, o32)["trajectory"][1].astype(np.float32)
    rng = np.random.default_rng(d["seed"])
    for k in range(6):
        got = sw.one((sw.perturb(X, rng),))
        assert np.allclose(got[:3], d["samples"][k], atol=2e-6), k
