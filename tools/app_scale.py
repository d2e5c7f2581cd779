#!/usr/bin/env python3
"""Times the C++ vo_complete counterpart (facade over the C ABI, host vectors, file I/O, map upkeep) on a
synthetic dataset in the reference's format.  usage (GPU box): python tools/app_scale.py [frames] [points]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
d = tempfile.mkdtemp(prefix="seq_")
out = tempfile.mkdtemp(prefix="out_")
t = time.time(); seq = vo.synth.sequence(seed=3000, n_frames=F, n_visible=N); vo.synth.write_sequence(seq, d)
print("dataset: %d frames x ~%d points written in %.1f s" % (F, N, time.time() - t), flush=True)
for extra in ([], ["--resident"]):
    t = time.time()
    r = subprocess.run([os.path.join(ROOT, "apps/bin/vo_complete"), d, out, "100"] + extra, capture_output=True, text=True)
    dt = time.time() - t
    print("vo_complete %s: rc %d, %.2f s total (file parsing included), %.1f ms per frame" % (" ".join(extra) or "(frame by frame)", r.returncode, dt, dt * 1e3 / F), flush=True)
    print(r.stdout.splitlines()[-1] if r.stdout else r.stderr[-500:])
t = time.time()
e = subprocess.run([os.path.join(ROOT, "apps/bin/evaluate"), d, out], capture_output=True, text=True)
print("evaluate: %.2f s" % (time.time() - t)); print(e.stdout)
