#!/usr/bin/env python3
"""Free device memory after repeated create/use/destroy cycles, by feature (diagnostic)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
vo = g.load_package()
hip = C.CDLL("libamdhip64.so")
def free_bytes():
    free, total = C.c_size_t(), C.c_size_t()
    hip.hipDeviceSynchronize(); hip.hipMemGetInfo(C.byref(free), C.byref(total))
    return free.value
fp = vo.synth.frame_pair(3000, seed=901)
def cycle(kind):
    c = vo.Context(0)
    if kind in ("frame", "capture", "all"):
        p = vo.FramePipeline(c, fp, n_iters=6)
        p.frame()
        if kind in ("capture", "all"):
            p.capture_frame(); p.frame_graph()
        p.counts(); p.close()
    if kind in ("match", "all"):
        for mode in (1, 2, 3):
            c.lib.vo_match_set_mode(c.h, mode)
            vo.compute_correspondences_images(fp["ref_app"], fp["cur_app"], ctx=c)
    if kind in ("event", "all"):
        e = vo.Event(c); e.record(c); e.wait(c); e.close()
    c.synchronize(); c.close()
for kind in ("ctx", "event", "match", "frame", "capture", "all"):
    cycle(kind)
    f0 = free_bytes()
    vals = []
    for _ in range(16):
        cycle(kind); vals.append((f0 - free_bytes()) / 2**20)
    print(f"{kind:8s}: drift MiB after each cycle: " + " ".join(f"{v:7.2f}" for v in vals), flush=True)
