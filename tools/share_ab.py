#!/usr/bin/env python3
"""Batched solver with fewer problems than CUs: picp_batch_kernel (one workgroup per problem, the other CUs idle) against
picp_batch_shared_kernel (the other CUs take trips off the problems' workgroups), which the environment switches
(VO_PICP_SHARE, read once per process: every setting runs in a child process).
  usage (GPU box): tools/share_ab.py [P,P,...] [setting setting ...]     setting: share=0 | keep=K (trips a home keeps) | g=G (wave-trips per chunk) | slack=S | absent=1, joined by '+'
Per setting and problem count: ms per vo_picp_solve_batch_dev call (form 2, 50 rounds, 50k correspondences per problem, the
gather pass included), whether problems with the same data got the same bits, and the largest difference of a pose entry
against the first setting's."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def child(P, ragged):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import __graft_entry__ as g
    vo = g.load_package()
    ctx = vo.Context(0)
    N, ITERS = int(os.environ.get("N", "50000")), int(os.environ.get("ITERS", "50"))
    fp = vo.synth.frame_pair(N, seed=2000)
    corr = np.stack([fp["gt_matches"][:, 1], fp["gt_matches"][:, 0]], 1).astype(np.int32)
    K = np.ascontiguousarray(fp["K"].T.reshape(-1), np.float32)
    assert ctx.lib.vo_picp_batch_set_form(ctx.h, 2) == 0
    n = np.full(P, N, np.int32)
    if ragged:                                   # sizes from empty to full, repeating every 8 problems
        n = np.array([(N * ((p % 8) + 0)) // 7 for p in range(P)], np.int32)
    d_world = ctx.to_device(np.tile(fp["model"], (P, 1))); d_meas = ctx.to_device(np.tile(fp["cur_pts"], (P, 1)))
    d_pairs = ctx.to_device(np.tile(corr, (P, 1))); d_n = ctx.to_device(n)
    d_T = ctx.alloc(P * 64); d_S = ctx.alloc(P * 16)

    def run():
        rc = ctx.lib.vo_picp_solve_batch_dev(ctx.h, C.c_int(P), C.c_int(480), C.c_int(640), C.c_int(0), C.c_int(10),
                                             K.ctypes.data_as(C.c_void_p), C.c_float(10000.0), C.c_int(0), C.c_void_p(d_world),
                                             C.c_size_t(N), C.c_void_p(d_meas), C.c_size_t(N), C.c_void_p(d_pairs), C.c_size_t(N),
                                             C.c_void_p(d_n), None, C.c_int(ITERS), C.c_void_p(d_T), C.c_void_p(d_S))
        assert rc == 0, ctx.lib.vo_last_error()
    for _ in range(2):
        run()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    T = np.zeros((P, 16), np.float32); ctx.d2h(T, d_T)
    S = np.zeros((P, 4), np.float32); ctx.d2h(S, d_S)
    same = all(T[p].tobytes() == T[p % 8].tobytes() and S[p].tobytes() == S[p % 8].tobytes() for p in range(P)) if ragged else \
        all(T[p].tobytes() == T[0].tobytes() and S[p].tobytes() == S[0].tobytes() for p in range(P))
    np.save(os.path.join(OUT, "share_ab_T.npy"), np.concatenate([T, S], 1))
    print(json.dumps({"ms": round(ms, 4), "same_data_same_bits": bool(same), "finite": bool(np.isfinite(T).all())}), flush=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    Ps = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [200]
    settings = sys.argv[2:] or ["share=0", "share=1"]
    names = {"share": "VO_PICP_SHARE", "keep": "VO_PICP_HELP_KEEP", "g": "VO_PICP_HELP_G", "slack": "VO_PICP_HELP_SLACK", "absent": "VO_PICP_HELP_ABSENT"}
    for ragged in ([int(os.environ['RAGGED'])] if 'RAGGED' in os.environ else (0, 1)):
        for P in Ps:
            base = None
            for st in settings:
                env = dict(os.environ)
                for kv in st.split("+"):
                    k, v = kv.split("=")
                    env[names[k]] = v
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(P), str(ragged)], env=env, capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    print("FAILED", st, P, r.stdout[-500:], r.stderr[-1500:], flush=True)
                    return 1
                res = json.loads(r.stdout.strip().splitlines()[-1])
                T = np.load(os.path.join(OUT, "share_ab_T.npy"))
                if base is None:
                    base = T
                diff = float(np.nanmax(np.abs(T[:, :16] - base[:, :16])))
                inl = int(np.abs(T[:, 18] - base[:, 18]).max())
                print(f"{'ragged ' if ragged else 'uniform'} P={P:4d} {st:24s} {res['ms']:8.3f} ms  same-data-same-bits {res['same_data_same_bits']}  "
                      f"max |dT| vs first {diff:.2e}  max |d inliers| {inl}", flush=True)
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        sys.exit(main())
