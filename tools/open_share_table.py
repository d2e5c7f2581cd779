#!/usr/bin/env python3
"""matcher_ms_by_open_share of one bench.py run (frame leg only): python tools/open_share_table.py [bench args]"""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--legs", "frame", "--strong-pairs", "0", "--steps", "20", "--warmup", "2"] + sys.argv[1:],
                   capture_output=True, text=True)
d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["batched_frames"]
t = d["matcher_ms_by_open_share"]
print("ms_per_batch %.3f  matcher by open share: " % d["ms_per_batch"] + "  ".join(f"{k}: {v:.3f}" for k, v in t.items() if k not in ("note", "whole_call_ms")))
print("whole call (event time): " + "  ".join(f"{k}: {v:.3f}" for k, v in t["whole_call_ms"].items()) + "   frames/s without copies %.0f" % d["frames_per_sec_without_copies"])
