#!/usr/bin/env python3
"""Frame-by-frame comparison of vo_complete on the example data in the default arithmetic against the reference-order run (--exact):
counts (matches, joined, inliers) and the largest pose difference per frame.  usage (GPU box, repo root): tools/cmp_fast_exact.py"""
import os, re, subprocess, sys, numpy as np
ROOT = os.getcwd()
BIN = os.path.join(ROOT, "apps", "bin"); DATA = os.path.join(ROOT, "tests", "golden", "example_data", "data")
def run(out, *flags):
    os.makedirs(out, exist_ok=True)
    r = subprocess.run([os.path.join(BIN, "vo_complete"), DATA, out, *flags], capture_output=True, text=True, timeout=300)
    poses = np.loadtxt(os.path.join(out, "poses_raw.txt")).astype(np.float32).reshape(-1, 4, 4)
    counts = np.array(re.findall(r"^meas-\d+\.dat: (\d+) matches, (\d+) model correspondences, (\d+) inliers", r.stdout, flags=re.M), dtype=int)
    return poses, counts
f, cf = run("/tmp/o_fast/"); e, ce = run("/tmp/o_exact/", "--exact")
d = np.abs(f - e).reshape(len(f), -1).max(1)
for t in range(len(cf)):
    if (cf[t] != ce[t]).any() or t < 45:
        print(t, cf[t], ce[t], "pose diff %.2e" % d[t + 2])
