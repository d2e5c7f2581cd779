#!/usr/bin/env python3
"""Aggregate frames/s when S independent frame pipelines run on S HIP streams of one GPU
(throughput mode of config 4: independent pairs).  usage: python tools/multistream_frames.py [S ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

vo = g.load_package()
n = 50000
frames = 40
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    ctxs = [vo.Context(0) for _ in range(S)]
    pipes = [vo.FramePipeline(c, vo.synth.frame_pair(n, seed=2000 + i), n_iters=50) for i, c in enumerate(ctxs)]
    use_graph = os.environ.get("VO_FRAME_GRAPH", "1") != "0"
    for p in pipes:
        p.capture_frame() if use_graph else p.frame()
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        for p in pipes:
            p.frame_graph() if use_graph else p.frame()
    for c in ctxs:
        c.synchronize()
    dt = time.perf_counter() - t0
    ok = all(np.abs(p.pose() - vo.synth.frame_pair(n, seed=2000 + i)["X_gt"]).max() < 1e-3 for i, p in enumerate(pipes))
    print(f"streams={S}: {S * frames / dt:8.1f} frames/s  ({dt * 1e3 / frames:.3f} ms per round of {S} frames)  poses ok={ok}")
    for p in pipes:
        p.close()
    for c in ctxs:
        c.close()
