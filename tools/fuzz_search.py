#!/usr/bin/env python3
"""Randomised campaign for the searches beside the matcher: the kd-tree's approximate modes (bestMatchFast / fastSearch: tree, leaves
and answers equal to the oracle's tree index for index, leaf order included) and the exact radius search (fullSearch) against the
oracle's double loop, on random set sizes, leaf sizes, radii and distributions.  usage (GPU box): tools/fuzz_search.py [seed] [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
from oracle.oracle import Oracle
o32 = Oracle(32)
ctx = vo.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 120)
n_case = fails = 0
while time.time() < t_end:
    nt = int(rng.choice([rng.integers(1, 60), rng.integers(60, 2000), rng.integers(2000, 12000)])); nq = int(rng.integers(1, 3000))
    kind = int(rng.integers(0, 4))
    if kind == 0: t = rng.uniform(-1, 1, (nt, 10))
    elif kind == 1: t = rng.normal(0, 0.05, (nt, 10)) + rng.integers(-1, 2, (nt, 10)) * 0.5           # clusters
    elif kind == 2: t = rng.integers(-3, 4, (nt, 10)) / 16.0                                          # lattice: ties, duplicates
    else:
        t = np.zeros((nt, 10)); t[:, int(rng.integers(0, 10))] = rng.uniform(-5, 5, nt); t += rng.normal(0, 1e-3, (nt, 10))
    t = t.astype(np.float32)
    src = t[rng.integers(0, nt, nq)].astype(np.float64)
    q = (src + rng.normal(0, float(rng.choice([0.0, 0.01, 0.05])), (nq, 10))).astype(np.float32)
    radius = float(rng.choice([0.1, 0.0625, 0.02, 0.4])); leaf = int(rng.choice([1, 2, 5, 10, 20, 50]))
    ok = True
    best_o, lists_o, nodes_o = o32.kdtree_fast(t, q, radius, leaf)
    kd = vo.KdTree(t, leaf, ctx=ctx)
    n, nodes, leaves = kd.info()
    best = kd.bestMatchFast(q, radius); lists = kd.fastSearch(q, radius)
    kd.close()
    if not (n == nt and nodes == nodes_o and np.array_equal(best, best_o) and len(lists) == len(lists_o) and all(np.array_equal(a, b) for a, b in zip(lists, lists_o))):
        ok = False; print("KDTREE FAIL", nt, nq, kind, radius, leaf)
    if nt * nq < 4e6:
        full = vo.radius_search(t, q, radius, ctx=ctx)
        full_o = o32.radius_search(t, q, radius, brute=True)
        if not (len(full) == len(full_o) and all(np.array_equal(a, b) for a, b in zip(full, full_o))):
            ok = False; print("RADIUS FAIL", nt, nq, kind, radius)
        if not all(set(a.tolist()) <= set(b.tolist()) for a, b in zip(lists, full)):
            ok = False; print("SUBSET FAIL", nt, nq, kind, radius, leaf)
    n_case += 1; fails += 0 if ok else 1
print("cases", n_case, "failures", fails)
