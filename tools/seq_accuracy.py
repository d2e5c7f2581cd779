#!/usr/bin/env python3
"""Accuracy of the config-3 chain against the generator's ground truth in the default arithmetic and in reference-order
arithmetic (--exact): the same measures the bench line reports (bench._sequence_metrics), side by side, so that a change of
the default mode's rounding can be told from a change of its accuracy.
usage (GPU box, repo root): tools/seq_accuracy.py [frames=200] [points=50000] [rounds=100]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
import bench
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ctx = vo.Context(0)
seq = vo.synth.sequence(seed=3000, n_frames=F, n_visible=N)
out = {}
for name, exact in (("default", False), ("reference-order", True)):
    sp = vo.SequencePipeline(ctx, seq, n_iters=R, exact=exact)
    sp.run()
    traj, counts = sp.trajectory(), sp.counts()
    sp.close()
    m = bench._sequence_metrics(vo, seq, traj)
    out[name] = (traj, counts)
    print("%-16s rmse_position %.3e  scale drift %.3e  mean / max orientation error %.2e / %.2e" %
          (name, m["rmse_position"], m["scale_ratio_drift"], m["mean_orientation_error"], m["max_orientation_error"]))
d = np.abs(out["default"][0] - out["reference-order"][0]).reshape(F, -1).max(1)
print("largest pose difference default vs reference-order: %.2e (frame %d); joined counts differ in %d frames (max %d)" %
      (d.max(), int(d.argmax()), int((out["default"][1][:, 1] != out["reference-order"][1][:, 1]).sum()),
       int(np.abs(out["default"][1][:, 1] - out["reference-order"][1][:, 1]).max())))
