#!/usr/bin/env python3
"""bench.py -- PICP iterations/sec at 50 000 correspondences on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no RANK in the environment the script starts the N ranks itself (a child
`python -m torch.distributed.run`, before anything here touches the GPU) and relays rank 0's line.

One STEP = one pass of the hot path over one frame pair already resident in
HBM: the solver is reset to the identity pose, the matched correspondences
(device index pairs produced by the matcher+join kernels before the timed
region) are gathered once (pack kernel) and `--iters` (50) Gauss-Newton rounds
run with no host round trip (BASELINE config 2).  value = steps*iters*N / time.
With N > 1 every rank owns its own frame pair (weak scaling, no data-path
collective) and the timed region ends with one RCCL all-gather of the poses.

The same JSON line also reports, outside the headline number: frames/sec of the
whole frame (match+join+transform+PICP+triangulate), the batched solver
(config 4's per-GPU share), the HBM roofline of the dominant kernel and the
CPU baseline (the oracle's scalar float32 restatement, 1 thread, rank 0 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
from tools import stamp as _stamp  # noqa: E402   (which sources a file under profiles/ was taken on)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_SUSTAINED_GBS = 6300.0     # MI355X_MICROARCH.md: what HBM sustains (float4 copy; 6.0-6.1 TB/s for a 1.2 GB in-order sweep)
BYTES_PER_CORR_ITER = 20       # world xyz 12 B + measurement uv 8 B


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch-pairs", type=int, default=200, help="problems of the batched-solver leg (0: skip)")
    ap.add_argument("--frame-steps", type=int, default=100, help="frames of the whole-frame leg (0: skip)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU-baseline leg (0: skip)")
    ap.add_argument("--no-extras", action="store_true", help="headline measurement only")
    ap.add_argument("--legs", default=None,
                    help="extra legs to run (comma list of frame,batched,sequence,cpu,api); default: all five on one "
                         "GPU, only `frame` (the sharded config-4 leg) when several ranks run")
    ap.add_argument("--seq-frames", type=int, default=200, help="frames of the sequence leg (config 3)")
    ap.add_argument("--seq-points", type=int, default=50000, help="landmarks in view per frame in the sequence leg")
    ap.add_argument("--seq-iters", type=int, default=100, help="PICP rounds per frame in the sequence leg (vo_complete.cpp:163)")
    ap.add_argument("--strong-pairs", type=int, default=1600,
                    help="frame pairs of the strong-scaling config-4 leg, sharded over the ranks (0: skip)")
    ap.add_argument("--strong-per-call", type=int, default=1600, help="frames per vo_frames_batch_dev call in that leg")
    ap.add_argument("--open-shares", type=str, default="0.01,0.05,0.25,1.0",
                    help="matcher stage of the 200-frame batch timed again with these shares of every frame's queries displaced "
                         "(no bitwise copy in the tree); empty: skip (the counter passes of tools/collect_profiles.sh do)")
    ap.add_argument("--gen-workers", type=int, default=None, help="host processes generating the config-4 pairs (default: CPU share, <= 16)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as a CHILD process group (never an exec of a
    process that has touched the GPU; nothing has, at this point) and exit with its return code.  Rank 0 of the
    children prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    import torch                                     # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    if os.environ.get("VO_BENCH_SHARE_GPU") == "1":  # rehearsal: all ranks on device 0 (see main)
        have = max(have, args.gpus) if have > 0 else 0
    if have < args.gpus:
        print(f"bench.py --gpus {args.gpus} needs {args.gpus} GPUs on this node, found {have}", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if 0 < args.strong_pairs < args.gpus:
        raise SystemExit(f"--strong-pairs {args.strong_pairs} < --gpus {args.gpus}: every rank needs at least one pair")
    # VO_BENCH_FORCE_LAUNCH=1: take the self-launch route with one rank too (rehearsal of the N-rank path on a 1-GPU box)
    if (args.gpus > 1 or os.environ.get("VO_BENCH_FORCE_LAUNCH") == "1") and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    vo = graft.load_package()
    from importlib import import_module
    vdist = import_module("visual_odometry_amd.dist")
    rank, local_rank, world = vdist.env_rank_world()
    # host processes that generate the synthetic pairs of the config-4 legs: forked now, while this process is
    # still clean (no torch, no HIP call)
    want_legs = set() if args.no_extras else set((args.legs or ("frame,batched,sequence,cpu,api" if world == 1 else "frame")).split(","))
    if "frame" in want_legs and args.frame_steps > 0:
        _PairGen.start_pool(world, args.gen_workers)
    import torch
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks (or plain `python bench.py --gpus N`)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # REHEARSAL of the N-rank path on a box with ONE GPU (VO_BENCH_SHARE_GPU=1): every rank drives device 0 and the collectives
    # go through gloo (RCCL refuses two ranks on one device; dist.gather_poses stages through the host then).  Sharding,
    # padding, gathers and slice checks are exactly the N-GPU code; the rates are NOT scaling numbers (the ranks share a chip)
    # and the line says so.
    share_gpu = os.environ.get("VO_BENCH_SHARE_GPU") == "1"
    gpu_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    # under torch.distributed.run (RANK set) the RCCL path runs even with one rank, so that a
    # 1-GPU box exercises exactly the code the N-GPU launch uses
    dist = vdist.init("gloo" if share_gpu else "nccl", gpu_index) if (world > 1 or "RANK" in os.environ) else None

    stream = torch.cuda.Stream(device=dev)
    ctx = vo.Context(gpu_index, stream.cuda_stream)
    dev_name, n_cu = ctx.device_info()
    lib = ctx.lib

    fp = vo.synth.frame_pair(args.points, seed=2000 + rank)
    pipe = vo.FramePipeline(ctx, fp, n_iters=args.iters, kernel_threshold=10000.0)
    pose_t = torch.zeros((1, 16), dtype=torch.float32, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        # inputs of the timed region: correspondences (cur_idx, model_idx) in device memory
        pipe.match(); pipe.join(); pipe.transform()
        ctx.synchronize()
        n_match, n_join, _ = pipe.counts().tolist()
        assert n_join == args.points, (n_match, n_join)

        def step():
            pipe.picp()

        # clocks, caches and the solver's graph settle within the first tens of milliseconds of work: a fixed pre-warm in
        # front of the W warm-up steps the caller asked for, so that a short run (--steps 20 --warmup 2) times the same steady
        # state as a long one (1.8 % apart without it)
        for _ in range(PREWARM_STEPS):
            step()
        for _ in range(args.warmup):
            step()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)          # device time of the solver launches alone
        if dist is not None:        # final exchange: poses of all ranks, RCCL over xGMI
            _chk(lib, lib.vo_picp_get_pose_dev(pipe.solver, C.c_void_p(pose_t.data_ptr())))
            all_poses = vdist.gather_poses(pose_t)
        barrier()
        t1 = time.perf_counter()
        elapsed = t1 - t0
        if dist is not None:
            elapsed = vdist.max_over_ranks(elapsed, dev)
        dev_ms = ev0.elapsed_time(ev1)

    # the work was real: the pose must be the ground truth of this rank's pair
    T = pipe.pose()
    pose_err = float(np.abs(T - fp["X_gt"]).max())
    chi_in, chi_out, n_in = pipe.stats()
    assert n_in == args.points and pose_err < 1e-3, (n_in, pose_err)
    if dist is not None:
        assert all_poses.shape == (world, 16)
        assert np.allclose(all_poses[rank].cpu().numpy().reshape(4, 4).T, T)

    total_iters = args.steps * args.iters * world
    value = total_iters / elapsed
    per_round_us = dev_ms * 1e3 / (args.steps * args.iters)
    alg_bytes = BYTES_PER_CORR_ITER * args.points
    achieved = alg_bytes / (per_round_us * 1e-6) / 1e9
    out = {
        **({"rehearsal": "VO_BENCH_SHARE_GPU=1: all ranks on ONE GPU, gloo collectives -- not a scaling measurement"} if share_gpu else {}),
        "metric": "PICP iterations/sec @50k pts",
        "evidence_stamp": _stamp.current(),
        "value": value, "unit": "iter/s", "n_gpus": world, "ranks_seen": (dist.get_world_size() if dist is not None else 1),
        "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM_STEPS,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: single frame pair per GPU, {args.points} correspondences, "
                               f"{args.iters} PICP rounds per step (pack + {args.iters} linearize/solve launches), "
                               "pose reset to identity each step; inputs resident in HBM",
                   "points": args.points, "iters_per_step": args.iters, "parallelism": f"pairs x{world}",
                   "device": dev_name, "compute_units": n_cu},
        "pose_err_vs_gt": pose_err, "num_inliers": n_in,
        "roofline": {"bound": "hbm", "kernel": "picp_round_kernel<true,false,true,false>", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": _pmc_traffic("vo::picp_round_kernel<true, false, true, false>", 256 * ((args.points + 255) // 256))[0],
                     "traffic_note": _pmc_traffic("vo::picp_round_kernel<true, false, true, false>", 256 * ((args.points + 255) // 256))[1],
                     "algorithmic_bytes_per_launch": alg_bytes, "launch_us": per_round_us,
                     "in_kernel": _headline_stamps(per_round_us),
                     "note": "one launch = one Gauss-Newton round over one 50k pair (1.0 MB, L2-resident): "
                             "latency-bound by the serial solve->linearize dependency, not by HBM; launch_us is "
                             "event time over the timed region / launches, i.e. it includes the kernel boundary "
                             "(and 1/50 of the step's gather and finishing launches); in_kernel: the phase stamps of the "
                             "same geometry from the diagnostic build (tools/stamps_headline.sh), kept under profiles/"},
    }

    # the rank-0-only legs (batched solver sweep, serial sequence, CPU baseline) would keep the other ranks
    # spinning in the final barrier: with several ranks only the sharded leg runs unless asked for explicitly
    default_legs = "frame,batched,sequence,cpu,api" if world == 1 else "frame"
    legs = set() if args.no_extras else set((args.legs or default_legs).split(","))
    if "frame" in legs and args.frame_steps > 0:           # every rank: config 4 (sharded pairs + pose gather)
        with torch.cuda.stream(stream):
            cfg4 = frame_throughput(vo, torch, ctx, stream, args, dist, vdist, rank, world)
        out["batched_frames"] = cfg4
        if args.strong_pairs > 0:
            with torch.cuda.stream(stream):
                out["batched_frames_strong"] = frame_throughput_strong(vo, torch, ctx, stream, args, dist, vdist, rank, world)
    if rank == 0 and legs:
        with torch.cuda.stream(stream):
            if args.frame_steps > 0 and "frame" in legs:
                out["frame"] = frame_leg(torch, ctx, stream, pipe, fp, args, vo)
                out["exact_mode"] = exact_leg(torch, ctx, stream, pipe, fp, args)
            if "api" in legs:
                out["api_one_round"] = api_leg(vo, ctx, pipe, fp, args, value)
            if args.batch_pairs > 0 and "batched" in legs:
                out["batched"] = batched_leg(torch, vo, ctx, stream, args)
            if args.seq_frames >= 3 and "sequence" in legs:
                out["sequence"] = sequence_leg(vo, ctx, args)
        if args.cpu_seconds > 0 and world == 1 and "cpu" in legs:
            out["cpu_baseline"] = cpu_leg(fp, pipe, args)
        _relate_exact_and_cpu(out)
    pipe.close()
    _PairGen.stop_pool()
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def _relate_exact_and_cpu(out):
    """The exact-mode leg and the CPU leg each carry their final pose (an ndarray) under "_pose": compare them when both
    ran (same pair, same rounds: bit for bit), and take the arrays out of the dicts whichever legs ran -- the line must
    stay JSON-serialisable for every --legs selection."""
    ex, cpu = out.get("exact_mode"), out.get("cpu_baseline")
    ex_pose = ex.pop("_pose", None) if ex is not None else None
    cpu_pose = cpu.pop("_pose", None) if cpu is not None else None
    if ex_pose is not None and cpu_pose is not None:
        ex["bit_identical_to_cpu_baseline"] = bool(np.array_equal(ex_pose, cpu_pose))
        ex["vs_cpu_baseline"] = ex["iters_per_sec"] / cpu["value"]


_CSRC_SHA = None


def _evidence_note(path):
    """how a committed evidence file relates to the tree this run executes: its stamp (tools/stamp.py) against the hash of
    the kernel sources at hand"""
    global _CSRC_SHA
    if _CSRC_SHA is None:
        _CSRC_SHA = _stamp.csrc_sha()
    if not path:
        return "no file"
    state, st = _stamp.status(path, _CSRC_SHA)
    name = os.path.basename(path)
    if state == "current":
        return f"{name}: taken at commit {st.get('commit')} on the kernel sources of this run (csrc_sha {_CSRC_SHA})"
    if state == "stale":
        return (f"{name}: STALE -- taken at commit {st.get('commit')} (csrc_sha {st.get('csrc_sha')}), the kernel sources of this run "
                f"hash to {_CSRC_SHA}")
    return f"{name}: unstamped (collected before round 5 introduced stamps); this run's csrc_sha is {_CSRC_SHA}"


def _pmc_file():
    """the newest committed PMC summary (profiles/rNN_pmc_fetch_write*.json)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_fetch_write*.json")))
    return files[-1] if files else None


def _headline_stamps(launch_us):
    """In-kernel phase split of the headline round (profiles/r04_headline_stamps.json: s_memtime stamps of workgroup 0 over
    1470 rounds of the timed geometry, VO_STAMPS build) beside this run's launch_us."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_headline_stamps.json")))
    try:
        path = files[-1]
        st = json.load(open(path))
    except Exception:
        return None
    return {"source": "profiles/" + os.path.basename(path), "evidence": _evidence_note(path), "rounds": st["rounds"],
            "in_kernel_us": st["in_kernel_us_mean"], "kernel_boundary_us": st["kernel_boundary_us_mean"],
            "round_to_round_us": st["round_to_round_us_mean"],
            "round_to_round_over_this_runs_launch_us": st["round_to_round_us_mean"] / launch_us,
            "phases_us": st["phases_us_mean"]}


def _pmc_traffic(kernel, grid_threads):
    """HBM bytes per launch from the committed PMC summary (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    this very script, tools/collect_profiles.sh), for the launch geometry `grid_threads`.  Returns (bytes, note) or
    (None, reason)."""
    path = _pmc_file()
    try:
        k = json.load(open(path))["kernels"][kernel]
        g = k["by_grid"][str(grid_threads)]
        fetch_kb, write_kb = g["FETCH_SIZE_KB_max"], g["WRITE_SIZE_KB_max"]
    except (OSError, KeyError, ValueError, TypeError):
        return None, "no committed PMC summary for this kernel and launch geometry"
    wide = k.get("wide_16B_loads", False)
    b = (2.0 * fetch_kb if wide else fetch_kb) * 1024.0 + write_kb * 1024.0
    return b, ("FETCH_SIZE x2 (gfx950 counts half of 16-B/lane coalesced reads) + WRITE_SIZE" if wide else
               "FETCH_SIZE + WRITE_SIZE as reported (4-B/lane loads: uncalibrated width)") + ", from " + _evidence_note(path)


FP32_VECTOR_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: peak FP32 (vector)


def _pmc_valu(kernels, pick=max):
    """executed VALU wave-instructions per launch (SQ_INSTS_VALU) of the named kernels from the committed counter summary
    (profiles/rNN_pmc_valu.json: one rocprofv3 --pmc pass of this script, tools/collect_profiles.sh), summed; for a kernel
    that ran at several launch geometries `pick` (max / min) chooses the largest / smallest grid -- the 200-frame call /
    the single frame.  (None, None, reason) when missing."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_valu.json")))
    if not files:
        return None, None, "no committed VALU counter summary"
    try:
        ks = json.load(open(files[-1]))["kernels"]
        total, util_w = 0.0, 0.0
        for k in kernels:
            if k not in ks:                      # (a kernel that did nothing in the counter pass -- a skipped fallback -- has no row)
                continue
            g = ks[k]["by_grid"]
            e = g[pick(g, key=lambda x: int(x))]
            total += e["valu_insts_per_launch"]
            util_w += e["valu_insts_per_launch"] * e["lane_utilisation"]
        return total, util_w / max(total, 1.0), "SQ_INSTS_VALU from " + os.path.basename(files[-1])
    except (OSError, KeyError, ValueError, TypeError) as e:
        return None, None, f"counter summary lacks {e!r}"


def _valu_roofline(kernels, seconds, scope, pick=max):
    """the matcher is FP32-VALU-bound -- executed VALU wave-instructions x 64 lanes x 2 flop (each counted as
    an FMA) over the live-measured time, against the FP32 vector peak"""
    insts, util, note = _pmc_valu(kernels, pick)
    if insts is None:
        return {"bound": "valu", "achieved": None, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "note": note}
    tf = insts * 128.0 / seconds / 1e12
    return {"bound": "valu", "scope": scope, "kernels": list(kernels), "achieved": tf, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / FP32_VECTOR_PEAK_TFLOPS, "valu_wave_instructions": insts, "lane_utilisation": util,
            "note": "achieved = executed VALU wave-instructions (" + note + ") x 64 lanes x 2 flop / live time of the stage: the share "
                    "of the FP32 vector issue rate the instruction stream occupies (an instruction is counted as an FMA with all "
                    "lanes active; lane_utilisation says how many are).  The peak is one wave64 FMA per 2 cycles and SIMD; measured "
                    "issue of plain FP32 streams on this part is 2.5 (VOP2) to 3.3-3.8 (VOP3) cycles (tools/micro/valu_rate.hip), "
                    "i.e. a practical ceiling of 53-80 % of it"}


PREWARM_STEPS = 100        # untimed steps in front of the caller's warm-up (see main)
MATCHER_CHAIN = ("vo::hash_rows_kernel", "vo::hash_table_kernel<14>", "vo::hash_probe_kernel<14>", "vo::hash_open_kernel",
                 "vo::cell_bounds_kernel", "vo::open_collect_kernel", "vo::open_scan_kernel",
                 "vo::cell_place_kernel", "vo::cell_offsets_kernel", "vo::cell_fine_kernel", "vo::cell_search_kernel<0>",
                 "vo::match_count_kernel", "vo::match_scatter_kernel")


def _matcher_roofline(frames, points, seconds):
    """The matcher stage of the batched call since round 4: row hashes, hash tables in LDS, one lookup per query (match.hip
    "hash-first"); the kernels behind it (open-query route, cell-hash search) only touch frames with open queries (none in this workload).  It streams
    every row once and gathers one tree row per query: HBM-side algorithmic bytes over time, the counter traffic of its
    kernels beside it, and -- SURVEY 8(d) asked for the matcher's VALU share -- the executed VALU stream as before."""
    alg = FRAME_ALG_BYTES["match"] * (points / 50000.0) * frames
    out = {"bound": "hbm", "scope": f"matcher stage of {frames} frames (exact-duplicate pass + skipped cell-hash search + compaction), one call",
           "kernels": list(MATCHER_CHAIN), "achieved": alg / seconds / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": alg / seconds / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_call": alg, "traffic": None}
    try:
        c = json.load(open(_pmc_file()))["batched_frames_call"]
        f = sum(v for k, v in c["FETCH_SIZE_KB_by_kernel"].items() if k in MATCHER_CHAIN)
        w = sum(v for k, v in c["WRITE_SIZE_KB_by_kernel"].items() if k in MATCHER_CHAIN)
        out["traffic"] = (f + w) * 1024.0
        out["traffic_over_algorithmic"] = out["traffic"] / alg
        out["traffic_note"] = ("FETCH_SIZE + WRITE_SIZE of the stage's kernels as reported (8-byte coalesced and scattered loads: "
                               "uncalibrated width; 8-byte coalesced streams read as half their bytes in tools/pmc_match.sh), from "
                               + os.path.basename(_pmc_file()))
    except (OSError, KeyError, ValueError, TypeError):
        out["traffic_note"] = "no committed PMC summary of the batched call"
    out["valu"] = _valu_roofline(MATCHER_CHAIN, seconds, "the same kernels")
    return out


def _pmc_frames_call():
    """HBM bytes of one whole vo_frames_batch_dev call (200 x 50k), summed over its kernels, from the same summary"""
    try:
        c = json.load(open(_pmc_file()))["batched_frames_call"]
        return c["bytes_corrected"], c["note"] + ", from " + _evidence_note(_pmc_file())
    except (OSError, KeyError, ValueError, TypeError):
        return None, "no committed PMC summary of the batched call"


def _chk(lib, rc):
    if rc != 0:
        raise RuntimeError(lib.vo_last_error().decode())


FRAME_ALG_BYTES = {"match": 4.4e6, "join": 1.2e6, "transform": 1.2e6, "pack": 2.4e6, "rounds_each": 1.0e6, "triangulate_v3": 6.2e6}


def _frame_alg_bytes(points, iters):
    """SURVEY 8(d) algorithmic bytes of one frame (match 40(N1+N2)+8Nq, join 8(C+C)+8C, transform 24 B/pt, gather
    28+20 B, `iters` x 20 B, triangulate v3 124 B/pair), scaled from the 50k figures."""
    f = points / 50000.0
    b = FRAME_ALG_BYTES
    return f * (b["match"] + b["join"] + b["transform"] + b["pack"] + iters * b["rounds_each"] + b["triangulate_v3"])


def _gen_pair(a):
    """worker of the generator pool: one config-4 pair, seed 4000+p (plain numpy)"""
    points, p = a
    vo = graft.load_package()
    f = vo.synth.frame_pair(points, seed=4000 + p)
    return {k: f[k] for k in ("ref_app", "cur_app", "ref_pts", "cur_pts", "model", "model_pairs", "X_gt", "K", "rows", "cols",
                               "z_near", "z_far")}


class _PairGen:
    """gen(lo, hi) for BatchPipeline: the pairs first+lo .. first+hi-1 of BASELINE configs[3] (seeds 4000+p), generated
    by the pool of host processes forked at the start of main() (before anything touched the GPU; no exec involved)"""
    pool = None
    workers = 1

    @classmethod
    def start_pool(cls, world, want=None):
        import multiprocessing as mp
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cls.workers = max(1, min(16, cores // max(world, 1))) if want is None else max(1, want)
        if cls.workers > 1:
            cls.pool = mp.get_context("fork").Pool(cls.workers)

    @classmethod
    def stop_pool(cls):
        if cls.pool:
            cls.pool.close(); cls.pool.join(); cls.pool = None

    def __init__(self, points, first):
        self.points, self.first = points, first

    def __call__(self, lo, hi):
        jobs = [(self.points, self.first + p) for p in range(lo, hi)]
        return self.pool.map(_gen_pair, jobs, chunksize=1) if self.pool else [_gen_pair(j) for j in jobs]


def _timed_batches(torch, ctx, bp, poses_t, dist, vdist, reps):
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    all_poses = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        bp.run()
        if dist is not None:
            all_poses = vdist.gather_poses(poses_t)        # (world*n_local, 16), rank-major = global pair order
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt = vdist.max_over_ranks(dt, poses_t.device)
    return dt / reps, all_poses


def _per_rank_rows(torch, ctx, bp, poses_t, dist, vdist, units, reps=5):
    """Outside the timed region, for reading a scaling run: every rank's own compute time per pass (its calls alone,
    synchronised on its own stream, no collective) and the time of the pose gather alone -- so that the first run on a
    real node shows whether a lost rank, a slow GPU or the collective is what the whole-job figure pays for."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        bp.run()
    ctx.synchronize()
    compute = (time.perf_counter() - t0) / reps
    gather = 0.0
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            vdist.gather_poses(poses_t)
        torch.cuda.synchronize()
        gather = (time.perf_counter() - t0) / reps
        rows = vdist.gather_values([compute, gather, float(units)], poses_t.device)
    else:
        rows = [[compute, gather, float(units)]]
    return {"ranks_seen": len(rows), "units_per_rank": [int(r[2]) for r in rows],
            "compute_ms_per_rank": [r[0] * 1e3 for r in rows], "frames_per_sec_per_rank": [r[2] / r[0] for r in rows],
            "gather_ms_per_rank": [r[1] * 1e3 for r in rows],
            "note": "per rank, outside the timed region: its own calls per pass without any collective, and the all-gather of the "
                    "poses alone (after a barrier)"}


def _check_batches(bp, points):
    P = bp.poses()
    err = float(np.abs(P - bp.X_gt).max())
    c = bp.counts()
    assert err < 1e-3 and int(c[:2].min()) == points and int(c[2].min()) > 0.5 * points, \
        (err, c.min(axis=1).tolist(), c.argmin(axis=1).tolist())
    return err


def frame_throughput(vo, torch, ctx, stream, args, dist=None, vdist=None, rank=0, world=1, frames=200):
    """BASELINE configs[3], weak form: `frames` independent frame pairs PER GPU (pairs 200*rank .. of the 1600,
    seeds 4000+p) resident in HBM, processed by one vo_frames_batch_dev call (every stage one batched launch,
    frame = a grid dimension; batched solver); with several ranks the job ends with ONE RCCL all-gather of the
    poses.  Called by every rank; timing bracketed by barrier + synchronize, max over ranks."""
    gen = _PairGen(args.points, frames * rank)
    poses_t = torch.zeros((frames, 16), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    bp = vo.BatchPipeline(ctx, gen, n_iters=args.iters, poses_ptr=poses_t.data_ptr(), n_frames=frames, upload_block=50)
    bp.run()
    ctx.synchronize()
    # ten calls between the barriers: the bracket itself (barrier, synchronize, the host waking up) is ~0.3 ms -- over three calls
    # of 2.6 ms it read as 0.13 ms per call (event time 2.56 against 2.69 ms wall)
    sec, all_poses = _timed_batches(torch, ctx, bp, poses_t, dist, vdist, 10)
    if dist is not None:
        assert all_poses.shape == (world * frames, 16)
        assert torch.equal(all_poses[rank * frames:(rank + 1) * frames], poses_t)
    ms = sec * 1e3
    per_rank = _per_rank_rows(torch, ctx, bp, poses_t, dist, vdist, frames)
    err = _check_batches(bp, args.points)
    # the matcher chain of the same frames alone (vo_match_appearances_batch_dev: bounds, level 1, offsets, level 2, search, compaction)
    bp.match_only(); ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(3):
        bp.match_only()
    e1.record(stream)
    ctx.synchronize()
    match_ms = e0.elapsed_time(e1) / 3
    assert int(bp.counts()[0].min()) == args.points
    # ... and when a share of every frame's queries has NO bitwise copy in the tree (new landmarks of a tracking frame: the
    # exact-duplicate pass leaves them open; up to nq / 16 of them per frame have the tree streamed past them, more go to the
    # sorted search): the synthetic pairs of this leg have none, so the stage's time above is its best case
    by_share = None
    shares = sorted(float(x) for x in args.open_shares.split(",") if x.strip())
    if dist is None and (frames, args.points) == (200, 50000) and shares:
        by_share = {"0": match_ms}
        e0.record(stream)
        for _ in range(3):
            bp.run()
        e1.record(stream)
        ctx.synchronize()
        call_by_share = {"0": e0.elapsed_time(e1) / 3}        # (event time like the rest of this table: ms_per_batch above is wall time with the step's barriers)
        done = 0.0
        for share in shares:
            if share >= 1.0:
                bp.perturb_cur_app(1.0, seed=1000)                       # every row: no bitwise copy left in any frame
            else:
                bp.perturb_cur_app(share - done, seed=int(share * 1000))     # (rows drawn anew: the shares add up, a few rows twice)
            done = share
            for _ in range(3):          # steady state: the automatic matcher takes two synchronised calls to learn what the data holds
                bp.match_only(); ctx.synchronize()
            e0.record(stream)
            for _ in range(3):
                bp.match_only()
            e1.record(stream)
            ctx.synchronize()
            by_share[f"{share:g}"] = e0.elapsed_time(e1) / 3
            assert int(bp.counts()[0].min()) > 0.99 * args.points       # (displaced by sigma 0.005: still inside the radius 0.1)
            for _ in range(2):
                bp.run(); ctx.synchronize()                             # ... and the whole call on the same frames
            e0.record(stream)
            for _ in range(3):
                bp.run()
            e1.record(stream)
            ctx.synchronize()
            call_by_share[f"{share:g}"] = e0.elapsed_time(e1) / 3
        by_share["note"] = ("matcher stage alone, ms per 200 x 50k frames, by the share of every frame's queries displaced so that "
                            "they have no bitwise copy in the tree (still matched: by the search)")
        by_share["whole_call_ms"] = call_by_share
    bp.close()
    without = None
    if by_share is not None and "1" in by_share:
        # descriptors recomputed per frame: no query is a bitwise copy, every one is answered by the nearest-neighbour search
        m_alg = FRAME_ALG_BYTES["match"] * frames
        without = {"frames_per_sec": frames / (by_share["whole_call_ms"]["1"] * 1e-3), "ms_per_call_event_time": by_share["whole_call_ms"]["1"],
                   "matcher_ms": by_share["1"],
                   "matcher_roofline": {"bound": "hbm", "scope": f"matcher stage of {frames} frames, no bitwise copies: grid bounds, "
                                        "level 1 / offsets / level 2 of the cell sort, cell-hash search, compaction", "achieved":
                                        m_alg / (by_share["1"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": m_alg / (by_share["1"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_call": m_alg, "traffic": None,
                                        "traffic_note": "counter passes of this variant: tools/pmc_match.sh with OPENS=1.0 (profiles/)"},
                   "note": "every current descriptor displaced by N(0, 0.005) per component (all still matched, by the search): "
                           "what a front end that recomputes descriptors per frame gets; the headline frames_per_sec is for "
                           "descriptors copied bit for bit, as the reference's data and SURVEY 8(d)'s generator have them"}
    alg = _frame_alg_bytes(args.points, args.iters) * frames
    gbs = alg / sec / 1e9
    return {"frames_per_gpu": frames, "n_gpus": world, "frames_total": frames * world, "ms_per_batch": ms, "scaling": "weak",
            "frames_per_sec": frames * world / (ms * 1e-3), "us_per_frame_per_gpu": ms * 1e3 / frames, "pose_err_vs_gt": err,
            "seeds": f"4000+p, p = {frames * rank}..{frames * rank + frames - 1} on this rank",
            "per_rank": per_rank,
            "matcher_ms_per_batch": match_ms,
            "matcher_ms_by_open_share": by_share,
            "frames_per_sec_without_copies": without["frames_per_sec"] if without else None,
            "without_copies": without,
            "matcher_roofline": _matcher_roofline(frames, args.points, match_ms * 1e-3)
            if (frames, args.points) == (200, 50000) else None,
            "roofline": {"bound": "hbm", "scope": "whole frame (all stages of one vo_frames_batch_dev call)", "achieved": gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": _pmc_frames_call()[0] if (frames, args.points, args.iters) == (200, 50000, 50) else None,
                         "traffic_note": _pmc_frames_call()[1],
                         "algorithmic_bytes_per_call": alg,
                         "note": "SURVEY 8(d) frame bytes (match 4.4 + join 1.2 + transform 1.2 + gather 2.4 + rounds x 1.0 + "
                                 "triangulate-v3 6.2 MB at 50k) x frames / wall time of the call, per GPU"},
            "note": "match + join + transform + 50 rounds + triangulate for every frame, one vo_frames_batch_dev call per "
                    "rank" + ("; + one all_gather_into_tensor of the poses per batch" if dist is not None else "")}


def frame_throughput_strong(vo, torch, ctx, stream, args, dist=None, vdist=None, rank=0, world=1):
    """BASELINE configs[3] as stated: `--strong-pairs` (1600) independent frame pairs IN TOTAL, seeds 4000+p, pair p on
    rank floor(p / (P/R)) (dist.shard_range: contiguous blocks), one RCCL all-gather of all the poses at the end.
    Total work is fixed as the rank count grows: strong scaling.  One GPU runs all 1600 (in calls of
    --strong-per-call frames)."""
    P = args.strong_pairs
    plan = vdist.StrongPlan(P, rank, world)                 # contiguous blocks; the gather needs equal-sized ones (padding)
    lo, hi, n_local, blk = plan.lo, plan.hi, plan.n_local, plan.blk
    gen = _PairGen(args.points, lo)
    t0 = time.perf_counter()
    poses_t = torch.zeros((blk, 16), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    bp = vo.BatchPipeline(ctx, gen, n_iters=args.iters, poses_ptr=poses_t.data_ptr(), n_frames=n_local, upload_block=50,
                          frames_per_call=args.strong_per_call)
    t_setup = time.perf_counter() - t0
    bp.run()
    ctx.synchronize()
    sec, all_poses = _timed_batches(torch, ctx, bp, poses_t, dist, vdist, 4)
    if dist is not None:
        assert all_poses.shape == (world * blk, 16)
        r0, r1 = plan.own_rows()
        assert torch.equal(all_poses[r0:r1], poses_t[:n_local])
        assert plan.global_order(all_poses).shape == (P, 16)
    per_rank = _per_rank_rows(torch, ctx, bp, poses_t, dist, vdist, n_local, reps=2)
    err = _check_batches(bp, args.points)
    bp.close()
    alg = _frame_alg_bytes(args.points, args.iters) * P
    return {"pairs_total": P, "n_gpus": world, "pairs_this_rank": n_local, "frames_per_call": args.strong_per_call,
            "calls_per_pass": len(bp.calls), "scaling": "strong", "seconds_per_pass": sec, "frames_per_sec": P / sec,
            "us_per_frame": sec * 1e6 / P, "pose_err_vs_gt": err, "seeds": f"4000+p, p = {lo}..{hi - 1} on this rank",
            "setup_s": t_setup, "generator_workers": _PairGen.workers, "per_rank": per_rank,
            "roofline": {"bound": "hbm", "scope": "whole frame, all GPUs", "achieved": alg / sec / 1e9,
                         "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": alg / sec / 1e9 / (HBM_PEAK_GBS * world),
                         "algorithmic_bytes_per_pass": alg},
            "note": "all pairs distinct (seeds 4000+p), resident in HBM before the timed region; every pass = the rank's "
                    "share through vo_frames_batch_dev" + (" + one all_gather_into_tensor of the poses" if dist is not None else "")}


def exact_leg(torch, ctx, stream, pipe, fp, args):
    """The same rounds on the same pair with the solver in reference-order arithmetic (vo_picp_set_exact): one workgroup,
    the serial chain of one dependent float add per correspondence.  Its pose is compared BIT FOR BIT with the CPU baseline's
    in cpu_leg (same rounds, same input)."""
    lib = ctx.lib
    _chk(lib, lib.vo_picp_set_exact(pipe.solver, 1))
    reps = 5
    try:
        pipe.picp()
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            pipe.picp()
        e1.record(stream)
        ctx.synchronize()
        ms = e0.elapsed_time(e1) / reps
        T = pipe.pose()
    finally:
        _chk(lib, lib.vo_picp_set_exact(pipe.solver, 0))
    pipe.picp()                                      # leave the fast mode's pose behind, as the later legs expect
    ctx.synchronize()
    return {"iters_per_sec": args.iters / (ms * 1e-3), "us_per_round": ms * 1e3 / args.iters, "points": args.points,
            "pose_err_vs_gt": float(np.abs(T - fp["X_gt"]).max()), "_pose": T,
            "note": "reference-order arithmetic (terms unfused, H / b / chi summed sequentially in correspondence order, Eigen's "
                    "LDLT, double sin/cos): bit-identical to the float32 CPU restatement; floor = one dependent v_add_f32 "
                    "(7.5 cycles) per correspondence and round"}


def api_leg(vo, ctx, pipe, fp, args, headline):
    """The reference's own call pattern (vo_complete.cpp:161-168): `iters` x oneRound(host pairs), then camera().
    Here through the C ABI the reference would bind: per step vo_picp_set_pose(identity),
    `iters` x vo_picp_one_round on the SAME host array of 50k pairs (compared in full on every call), vo_picp_get_pose.
    `ctypes`: this process (every call pays Python's foreign-call overhead); `cpp`: apps/one_round_rate, the same loop on
    vo::PICPSolver::oneRound from C++ (a child process on the same GPU, run after this one's work has drained)."""
    import subprocess
    lib = ctx.lib
    pairs = np.ascontiguousarray(pipe.fetch("join"))
    n = len(pairs)
    s = vo.PICPSolver(ctx)
    s.setKernelThreshold(10000.0)
    s.init(vo.Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4), ctx=ctx), fp["model"], fp["cur_pts"])
    ident = np.ascontiguousarray(np.eye(4, dtype=np.float32))
    T = np.zeros(16, np.float32)
    one_round, set_pose, get_pose = lib.vo_picp_one_round, lib.vo_picp_set_pose, lib.vo_picp_get_pose
    a_h, a_p, a_n, a_k = s.h, C.c_void_p(pairs.ctypes.data), C.c_int(n), C.c_int(0)
    a_i, a_T = C.c_void_p(ident.ctypes.data), C.c_void_p(T.ctypes.data)

    def step():
        rc = set_pose(a_h, a_i)
        t0 = time.perf_counter()
        for _ in range(args.iters):
            rc |= one_round(a_h, a_p, a_n, a_k)
        t1 = time.perf_counter()
        rc |= get_pose(a_h, a_T)
        _chk(lib, rc)
        return t1 - t0

    steps = max(20, min(args.steps, 200))
    for _ in range(20):
        step()
    calls = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        calls += step()
    dt = time.perf_counter() - t0
    err = float(np.abs(T.reshape(4, 4).T - fp["X_gt"]).max())
    assert err < 1e-3 and s.numInliers() == n, (err, s.numInliers())
    open_rounds, spec, redone = s.chainInfo()
    s.close()
    out = {"what": f"per step: vo_picp_set_pose(identity), {args.iters} x vo_picp_one_round(host array of {n} pairs), vo_picp_get_pose "
                   "-- the reference's loop vo_complete.cpp:161-168; one kernel launch per call, the pairs compared in full on "
                   "every call while the round runs",
           "ctypes": {"iters_per_sec": steps * args.iters / dt, "us_per_round": dt * 1e6 / (steps * args.iters),
                      "host_us_per_call": calls * 1e6 / (steps * args.iters), "steps": steps, "pose_err_vs_gt": err,
                      "speculative_calls": spec, "repeated_calls": redone},
           "headline_iters_per_sec": headline}
    exe = os.path.join(ROOT, "apps", "bin", "one_round_rate")
    try:
        if not os.path.exists(exe):
            subprocess.run(["make", "-C", os.path.join(ROOT, "apps"), "-s", "bin/one_round_rate"], check=True, timeout=300)
        ctx.synchronize()
        r = subprocess.run([exe, str(args.points), str(args.iters), str(steps), "20"], capture_output=True, text=True, timeout=300)
        cpp = json.loads(r.stdout.strip().splitlines()[-1])
        cpp["returncode"] = r.returncode
        out["cpp"] = cpp
        out["iters_per_sec"] = cpp["loop"]["iters_per_sec"]
        out["host_us_per_call"] = cpp["loop"]["host_us_per_call"]
        out["vs_headline"] = cpp["loop"]["iters_per_sec"] / headline
    except Exception as e:      # (the C++ driver is an extra: the ctypes figures stand on their own)
        out["cpp"] = {"error": repr(e)}
        out["iters_per_sec"] = out["ctypes"]["iters_per_sec"]
        out["host_us_per_call"] = out["ctypes"]["host_us_per_call"]
        out["vs_headline"] = out["iters_per_sec"] / headline
    return out


def frame_leg(torch, ctx, stream, pipe, fp, args, vo_mod=None):
    """Whole frame: match + join + transform + PICP + triangulate, device-resident."""
    for _ in range(3):
        pipe.frame()
    ctx.synchronize()
    stages = {}
    lib = ctx.lib
    _chk(lib, lib.vo_match_set_mode(ctx.h, 1))       # every (query, tree point) pair
    for _ in range(2):
        pipe.match()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(args.frame_steps):
        pipe.match()
    e1.record(stream)
    ctx.synchronize()
    stages["match_full_scan_ms"] = e0.elapsed_time(e1) / args.frame_steps
    _chk(lib, lib.vo_match_set_mode(ctx.h, 0))       # default: bucket-pruned at this size
    for _ in range(2):
        pipe.match()
    for name, fn in (("match", pipe.match), ("join", pipe.join), ("transform", pipe.transform),
                     ("picp", pipe.picp), ("triangulate", pipe.triangulate)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.frame_steps):
            fn()
        e1.record(stream)
        ctx.synchronize()
        stages[name + "_ms"] = e0.elapsed_time(e1) / args.frame_steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.frame_steps):
        pipe.frame()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    c = pipe.counts().tolist()
    n1, n2 = pipe.n_ref, pipe.n_cur
    match_flops = 30.0 * n1 * n2                     # 30 flop per (tree, query) pair
    roofs = {}
    if (n1, n2) == (50000, 50000):
        roofs = {"match_full_scan_roofline": _valu_roofline(("vo::match_init_kernel", "vo::match_kernel<false>", "vo::match_count_kernel", "vo::match_scatter_kernel"),
                                                            stages["match_full_scan_ms"] * 1e-3, "one frame, every (query, tree point) pair visited", min),
                 "match_roofline": _valu_roofline(("vo::match_minmax_kernel", "vo::match_bucket_hist_kernel", "vo::match_bucket_offsets_kernel",
                                                   "vo::match_bucket_place_kernel", "vo::match_pruned_kernel", "vo::match_count_kernel",
                                                   "vo::match_scatter_kernel"), stages["match_ms"] * 1e-3, "one frame, bucket-pruned scan (default)", min)}
    return {"frames_per_sec": args.frame_steps / dt, "ms_per_frame": dt * 1e3 / args.frame_steps, 
            "counts": {"matches": c[0], "joined": c[1], "triangulated": c[2]}, **stages, **roofs,
            "match_full_scan_equiv_tflops": match_flops / (stages["match_full_scan_ms"] * 1e-3) / 1e12,
            "match_note": "match_ms: default (bucket-pruned exact scan); match_full_scan_ms: every pair visited, "
                          "bit-exact 3-term early exit; equiv_tflops = 30*N1*N2 flop / time (brute-force-equivalent "
                          "rate, not executed flops)"}


def sequence_leg(vo, ctx, args):
    """BASELINE configs[2]: a serial synthetic sequence (epipolar initialisation, then per frame
    match -> join -> transform -> seq_iters rounds -> triangulate), everything resident in HBM."""
    t0 = time.perf_counter()
    seq = vo.synth.sequence(seed=3000, n_frames=args.seq_frames, n_visible=args.seq_points)
    t_gen = time.perf_counter() - t0
    def timed(overlap, prematch=False, keep_map=False):
        sp = vo.SequencePipeline(ctx, seq, n_iters=args.seq_iters, overlap_match=overlap, prematch=prematch, keep_map=keep_map)
        sp.run(); ctx.synchronize()                  # warm-up pass: sizes every workspace, builds the solver graph
        inits = []
        for _ in range(3):                            # the first pair three times (start() begins the sequence anew): the median
            t0 = time.perf_counter()                  # (a single sample of this sub-millisecond host-driven step read 0.45 .. 1.2 ms over runs)
            sp.start(); ctx.synchronize()
            inits.append(time.perf_counter() - t0)
        t1 = time.perf_counter()
        t0 = t1 - sorted(inits)[1]
        for t in range(2, sp.F):
            sp.step(t)
        ctx.synchronize()
        t2 = time.perf_counter()
        res = [sp.trajectory(), sp.counts(), t1 - t0, t2 - t1, sp.F]
        if keep_map:
            res.append(len(sp.map))
        if prematch:                                  # the one batched matcher call on its own
            ctx.synchronize(); t3 = time.perf_counter()
            sp.match_all(); ctx.synchronize()
            res.append(time.perf_counter() - t3)
        sp.close()
        return res

    traj, counts, init_s, chain_s, F = timed(False)
    traj2, counts2, _, chain2, _ = timed(True)
    assert np.array_equal(traj, traj2) and np.array_equal(counts, counts2), "overlapped matcher changed the result"
    traj3, counts3, init3, chain3, _, match_all_s = timed(False, prematch=True)
    assert np.array_equal(traj, traj3) and np.array_equal(counts, counts3), "matching up front changed the result"
    # the loop body as the reference has it: map.update(history * triangulated_pc) and the history step inside the chain
    # (vo_complete.cpp:175-176), on the device
    traj4, counts4, init4, chain4, _, map_entries = timed(False, keep_map=True)
    assert np.array_equal(traj, traj4) and np.array_equal(counts, counts4), "the map upkeep changed the chain"
    m = _sequence_metrics(vo, seq, traj)
    n = [len(f["pts"]) for f in seq["frames"]]
    return {"frames": F, "points_per_frame": {"min": int(min(n)), "max": int(max(n))}, "landmarks": len(seq["world_xyz"]),
            "iters_per_frame": args.seq_iters, "init_ms": init_s * 1e3, "init_ms_note": "median of three runs of the first pair",
            "chain_ms": chain_s * 1e3, "frames_per_sec": (F - 2) / chain_s,
            "ms_per_frame": chain_s * 1e3 / (F - 2),
            "picp_iters_per_sec": (F - 2) * args.seq_iters / chain_s,
            "with_map": {"frames_per_sec": (F - 2) / chain4, "ms_per_frame": chain4 * 1e3 / (F - 2), "chain_ms": chain4 * 1e3,
                         "map_entries": map_entries, "map_update_ms_per_frame": (chain4 - chain_s) * 1e3 / (F - 2),
                         "note": "the same chain with the loop body's map upkeep inside it, on the device: map.update(history * "
                                 "triangulated_pc) keyed by exact appearance equality (PointCloud.h:52-66) + history = history * "
                                 "pose^-1 (vo_complete.cpp:175-176); frames_per_sec above is the chain without it"},
            "frames_per_sec_matcher_on_second_stream": (F - 2) / chain2,
            "matched_up_front": {"frames_per_sec": (F - 2) / (chain3 + match_all_s), "chain_ms": chain3 * 1e3,
                                 "match_all_ms": match_all_s * 1e3, "init_ms": init3 * 1e3,
                                 "note": "all F-1 pairs matched by one vo_match_appearances_batch_dev call at the start (the "
                                         "matcher needs the appearances alone; vo_complete has every measurement file on "
                                         "hand); the whole call is charged to the F-2 chained frames; identical results"},
            "joined_per_frame": {"min": int(counts[2:, 1].min()), "max": int(counts[2:, 1].max())},
            "accuracy_vs_ground_truth": m, "generate_s": t_gen,
            "note": "init = match + vo_estimate_transform (host 8-point, once) + triangulate of the first pair; "
                    "chain = frames 2.. enqueued back to back on one stream with no host synchronisation; "
                    "frames_per_sec_matcher_on_second_stream: the matcher of frame t+1 under the solver rounds of frame t "
                    "(vo_event_*; identical results) -- measured, not the default: its waves take issue slots from the "
                    "latency-bound rounds; "
                    "accuracy: relative poses against the generator's ground truth (evaluate.cpp's measures)"}


def _sequence_metrics(vo, seq, traj):
    """evaluate.cpp:18-86 on relative camera poses: rotation part of X_est^-1 X_gt, translation-norm
    ratio (its median fixes the monocular scale), position RMSE of the chained trajectory after scaling."""
    Xgt = vo.synth.sequence_gt_relative(seq)
    e_rot, ratio = [], []
    for t in range(1, len(traj)):
        Xe = traj[t].astype(np.float64)
        e_rot.append(float(np.trace(np.eye(3) - Xe[:3, :3].T @ Xgt[t - 1][:3, :3])))
        ratio.append(float(np.linalg.norm(Xe[:3, 3]) / np.linalg.norm(Xgt[t - 1][:3, 3])))
    r = float(np.median(ratio))
    We, Wg = np.eye(4), np.eye(4)
    err = []
    for t in range(1, len(traj)):
        We = We @ np.linalg.inv(traj[t].astype(np.float64)); Wg = Wg @ np.linalg.inv(Xgt[t - 1])
        err.append(np.linalg.norm(We[:3, 3] / r - Wg[:3, 3]) ** 2)
    return {"mean_orientation_error": float(np.mean(e_rot)), "max_orientation_error": float(np.max(np.abs(e_rot))),
            "scale_ratio_median": r, "scale_ratio_drift": float(max(ratio) / min(ratio) - 1.0),
            "rmse_position": float(np.sqrt(np.mean(err))), "path_length": float(seq["step"] * (len(traj) - 1))}


def batched_leg(torch, vo, ctx, stream, args):
    """config 4's per-GPU share (200 problems), plus the same kernel with the chip full (256, 512
    problems: one workgroup per problem, 256 CUs) to show where it saturates."""
    out = _batched_run(torch, vo, ctx, stream, args, args.batch_pairs)
    if args.batch_pairs == 200:
        out["chip_full"] = []
        for P in (256, 512, 4096):
            r = _batched_run(torch, vo, ctx, stream, args, P)
            ws = r["working_set_MB"] * 1e6
            tb = r["frac_traffic"] * HBM_PEAK_GBS / 1e3
            out["chip_full"].append({"pairs": P, "kernel_ms": r["kernel_ms"], "iters_per_sec": r["iters_per_sec"],
                                     "achieved_GBs": r["roofline"]["achieved"], "frac": r["roofline"]["frac"],
                                     "frac_algorithmic": r["frac_algorithmic"], "frac_traffic": r["frac_traffic"],
                                     "traffic_TBs": tb, "frac_traffic_of_sustained_6.3TBs": tb * 1e3 / HBM_SUSTAINED_GBS,
                                     "working_set_MB": r["working_set_MB"],
                                     "working_set_fits_infinity_cache": r["working_set_fits_infinity_cache"],
                                     "infinity_cache_share_upper_bound": min(1.0, 256 * 2 ** 20 / ws),
                                     "traffic_TBs_from_hbm_lower_bound": tb * (1.0 - min(1.0, 256 * 2 ** 20 / ws)),
                                     "served_by": ("Infinity Cache" if ws < 256 * 2 ** 20 else
                                                   "Infinity-Cache-assisted (working set < 8 x the 256 MiB cache)" if ws < 8 * 256 * 2 ** 20
                                                   else "HBM")})
        past = [c for c in out["chip_full"] if c["served_by"] == "HBM"]
        if past:      # the one point to read as an HBM fraction: 2 GiB of packed correspondences swept 50 times, 8 x the Infinity Cache
            out["hbm_point"] = dict(past[-1], note="working set 16 x the 256 MiB Infinity Cache, swept in order once per round: whatever the "
                                                    "cache's replacement policy, at most cache / working set = 6 % of the bytes can come from it "
                                                    "(infinity_cache_share_upper_bound), so at least traffic_TBs_from_hbm_lower_bound comes from "
                                                    "HBM; frac_traffic (bytes that leave the LDS, over time) is quoted against the 8.0 TB/s "
                                                    "specification and against the 6.3 TB/s the guide measures as sustained "
                                                    "(frac_traffic_of_sustained_6.3TBs).  The 256- and 512-problem points are "
                                                    "Infinity-Cache-assisted and are not HBM figures.")
        # a few problems per call: the launch-per-round form (problem = grid dimension) against one workgroup per problem
        out["few_problems"] = []
        for P in (4, 16):
            row = {"pairs": P}
            for form, name in ((1, "launch_per_round"), (2, "one_workgroup_per_problem"), (0, "auto")):
                _chk(ctx.lib, ctx.lib.vo_picp_batch_set_form(ctx.h, form))
                r = _batched_run(torch, vo, ctx, stream, args, P)
                row[name + "_ms"] = r["ms_per_call"]
            _chk(ctx.lib, ctx.lib.vo_picp_batch_set_form(ctx.h, 0))
            row["iters_per_sec"] = P * args.iters / (row["auto_ms"] * 1e-3)
            out["few_problems"].append(row)
        # fewer problems than CUs: the waves of the CUs without a problem take chunks of the others' rounds (picp_batch_shared_kernel,
        # up to 0.65 problems per CU; VO_PICP_SHARE is read per call) against one workgroup per problem alone
        out["helper_waves"] = []
        for P in (32, 64, 128, 160):
            row = {"pairs": P}
            for share, name in (("0", "alone"), ("1", "with_helpers")):
                os.environ["VO_PICP_SHARE"] = share
                try:
                    _chk(ctx.lib, ctx.lib.vo_picp_batch_set_form(ctx.h, 2))
                    r = _batched_run(torch, vo, ctx, stream, args, P)
                finally:
                    os.environ.pop("VO_PICP_SHARE", None)
                    _chk(ctx.lib, ctx.lib.vo_picp_batch_set_form(ctx.h, 0))
                row[name + "_ms"] = r["ms_per_call"]
            row["speedup"] = row["alone_ms"] / row["with_helpers_ms"]
            row["iters_per_sec"] = P * args.iters / (row["with_helpers_ms"] * 1e-3)
            out["helper_waves"].append(row)
    return out


def _batched_run(torch, vo, ctx, stream, args, P):
    n, iters = args.points, args.iters
    lib = ctx.lib
    distinct = min(P, 4)
    fps = [vo.synth.frame_pair(n, seed=4000 + p) for p in range(distinct)]
    pairs = []
    for f in fps:
        mp = np.full(len(f["ref_app"]), -1, np.int64)
        mp[f["model_pairs"][:, 0]] = f["model_pairs"][:, 1]
        gt = f["gt_matches"]
        pairs.append(np.stack([gt[:, 1], mp[gt[:, 0]]], axis=1).astype(np.int32))
    d_world = ctx.alloc(P * n * 12); d_meas = ctx.alloc(P * n * 8); d_pairs = ctx.alloc(P * n * 8)
    for p in range(P):      # distinct copies in HBM: the traffic is real even where the values repeat
        f = fps[p % distinct]
        ctx.h2d(d_world + p * n * 12, f["model"]); ctx.h2d(d_meas + p * n * 8, f["cur_pts"])
        ctx.h2d(d_pairs + p * n * 8, pairs[p % distinct])
    d_n = ctx.to_device(np.full(P, n, np.int32))
    d_T = ctx.alloc(P * 64); d_stats = ctx.alloc(P * 16)
    K = np.ascontiguousarray(fps[0]["K"].T).ravel()

    def run(n_it=iters):
        _chk(lib, lib.vo_picp_solve_batch_dev(ctx.h, P, 480, 640, 0, 10, K.ctypes.data_as(C.c_void_p),
                                              C.c_float(10000.0), 0, C.c_void_p(d_world), C.c_size_t(n),
                                              C.c_void_p(d_meas), C.c_size_t(n), C.c_void_p(d_pairs), C.c_size_t(n),
                                              C.c_void_p(d_n), None, n_it, C.c_void_p(d_T), C.c_void_p(d_stats)))

    def timed(n_it, reps):
        run(n_it); ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            run(n_it)
        e1.record(stream)
        ctx.synchronize()
        return e0.elapsed_time(e1) / reps
    pack_ms = timed(0, 5)          # gather pass + an iteration-less solver launch
    ms = timed(iters, 5)           # gather pass + all rounds
    kernel_ms = ms - pack_ms       # picp_batch_kernel alone (rocprofv3 average must agree)
    T = np.zeros((P, 16), np.float32); st = np.zeros((P, 4), np.float32)
    ctx.d2h(T, d_T); ctx.d2h(st, d_stats)
    err = max(float(np.abs(T[p].reshape(4, 4).T - fps[p % distinct]["X_gt"]).max()) for p in range(P))
    assert err < 1e-3 and np.all(st[:, 2] == n), (err, st[:, 2].min())
    for d in (d_world, d_meas, d_pairs, d_n, d_T, d_stats):
        ctx.free(d)
    # per call: one gather pass (8 B pair + 20 B point data read, 20 B written) + iters streaming passes of 20 B
    alg = P * n * BYTES_PER_CORR_ITER * iters
    gbs = alg / (kernel_ms * 1e-3) / 1e9
    # what of it can reach HBM at all: the first two trips of every thread (2 x 768 threads x 4 correspondences) stay in LDS
    # after round 0 (PICP_BATCH_LDS_TRIPS, picp.hip); and a working set below the 256 MiB Infinity Cache is re-read from it
    resident = min(2 * 768 * 4, n) / n * (iters - 1) / iters
    traffic_model = alg * (1.0 - resident)
    working_set = P * n * BYTES_PER_CORR_ITER
    return {"pairs": P, "ms_per_call": ms, "pack_ms": pack_ms, "kernel_ms": kernel_ms,
            "frac_algorithmic": gbs / HBM_PEAK_GBS,
            "frac_traffic": traffic_model / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "working_set_MB": working_set / 1e6, "working_set_fits_infinity_cache": bool(working_set < 256 * 2 ** 20),
            "frac_note": "frac_algorithmic = 20 B x correspondences x rounds / time / 8 TB/s; frac_traffic = the same without the "
                         "bytes that stay in LDS after round 0; a working set below 256 MiB is served (partly) by the Infinity "
                         "Cache, whose hits FETCH_SIZE counts -- the HBM figure is the chip_full point whose working set exceeds it",
            "iters_per_sec": P * iters / (ms * 1e-3),
            "pair_solves_per_sec": P / (ms * 1e-3), "pose_err_vs_gt": err,
            "roofline": {"bound": "hbm", "kernel": "picp_batch_kernel<true,false>", "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": _pmc_traffic("vo::picp_batch_kernel<true, false>", 768 * P)[0] if (n, iters) == (50000, 50) else None,
                         "traffic_note": _pmc_traffic("vo::picp_batch_kernel<true, false>", 768 * P)[1],
                         "algorithmic_bytes_per_launch": alg,
                         "launch_us": kernel_ms * 1e3,
                         "note": f"{P} x {n} x 20 B x {iters} rounds of algorithmic bytes over kernel_ms = ms_per_call - "
                                 "pack_ms (the gather pass is a separate kernel); iters_per_sec uses the whole call"}}


def _cgroup_cpu_quota():
    """CPU cores granted by the cgroup (v2 cpu.max, v1 cfs quota), or None when unlimited / unknown"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / p if q > 0 else None
    except (OSError, ValueError):
        return None


def _median_time(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def cpu_leg(fp, pipe, args):
    """The oracle's float32 restatement of PICPSolver::oneRound on the host, 1 thread (the reference has no
    threading), same 50k pair, same 50 rounds; median of >= 5 repetitions after one warm-up."""
    from oracle.oracle import Camera as OCam, Oracle, native_lib
    o = Oracle(32)
    corr = pipe.fetch("join")
    cam = OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4))
    res = {}

    def one(orc, key):
        res[key] = orc.picp_solve(cam, fp["model"], fp["cur_pts"], corr, args.iters, 10000.0, False, trace=False)
    # repetitions: at least 5, more while the budget lasts (one repetition = 50 rounds, ~60 ms)
    t_one, _ = _median_time(lambda: one(o, "st"), reps=3)
    reps = int(max(5, min(400, args.cpu_seconds / max(t_one, 1e-6))))
    t_med, reps = _median_time(lambda: one(o, "st"), reps=reps, warm=0)
    r = res["st"]
    # the strong baseline: the same loop over all the cores this job may use (per-thread partial sums)
    try:
        n_thr = len(os.sched_getaffinity(0))
    except AttributeError:
        n_thr = os.cpu_count() or 1
    n_aff = max(1, n_thr)
    # "all the cores this job may use" = the affinity mask capped by the cgroup's CPU quota: a 1-GPU box shows 256 cores in
    # the mask and grants about 16 of them (cpu.max) -- 256 OpenMP threads on a 16-core quota do not finish a round
    quota = _cgroup_cpu_quota()
    n_use = max(1, min(n_aff, int(quota + 0.5))) if quota else min(n_aff, 16)
    sweep = []
    counts = sorted({t for t in (16, n_use) if t <= n_use} or {n_use})
    for n_thr in counts:
        def mt():
            res["mt"] = o.picp_solve_mt(cam, fp["model"], fp["cur_pts"], corr, args.iters, n_thr, 10000.0)
        budget = 0.3 * args.cpu_seconds / len(counts)
        t_mt, mt_reps = _median_time(mt, reps=int(max(5, min(200, budget / max(t_one / min(n_thr, 16), 1e-6)))))
        rm = res["mt"]
        sweep.append({"value": args.iters / t_mt, "unit": "iter/s", "cores": rm["threads"], "repetitions": mt_reps,
                      "pose_diff_vs_single_thread": float(np.abs(rm["T"] - r["T"]).max())})
    all_cores = dict(sweep[-1], kind="port, OpenMP over contiguous chunks with per-thread H/b partials (the reference itself has "
                                     "no threading); cores = every core this job may use: affinity mask capped by the cgroup CPU quota",
                     affinity_mask_cores=n_aff, cgroup_cpu_quota=quota,
                     threads_sweep=[{"threads": e["cores"], "value": e["value"]} for e in sweep],
                     best={"threads": max(sweep, key=lambda e: e["value"])["cores"], "value": max(e["value"] for e in sweep)})
    # separately labelled: the same C sources compiled -march=native on this host
    try:
        on = Oracle(32, library=native_lib())
        t_nat, nat_reps = _median_time(lambda: one(on, "nat"), reps=max(5, reps // 4))
        march_native = {"value": args.iters / t_nat, "unit": "iter/s", "cores": 1, "repetitions": nat_reps,
                        "kind": "port, gcc -O3 -march=native -ffp-contract=off (built on this host)",
                        "pose_equal_to_baseline_build": bool(np.array_equal(res["nat"]["T"], r["T"]))}
    except Exception as e:          # no compiler on the host: report, do not fail the bench
        march_native = {"value": None, "error": repr(e)}
    gpu_T = pipe.pose()
    # other stages of the frame on the host: medians of 5 repetitions; bounded samples where the
    # reference's own form is quadratic
    m = pipe.fetch("match")
    t_tri, _ = _median_time(lambda: o.triangulate(fp["K"], gpu_T, m, fp["ref_pts"], fp["cur_pts"]))
    t_join, _ = _median_time(lambda: o.join(m, fp["model_pairs"], linear=True))
    nj_s = min(2500, len(m))
    t_jq, _ = _median_time(lambda: o.join(m[:nj_s], fp["model_pairs"]), reps=5)
    jq_equal = bool(np.array_equal(o.join(m[:nj_s], fp["model_pairs"]), o.join(m[:nj_s], fp["model_pairs"], linear=True)))
    nq_s = min(100, len(fp["cur_app"]))
    t_match, _ = _median_time(lambda: o.match(fp["ref_app"], fp["cur_app"][:nq_s]), reps=5)
    kd = []
    for _ in range(5):
        mk, t_build, t_query = o.match_kdtree(fp["ref_app"], fp["cur_app"], timing=True)
        kd.append((t_build + t_query, t_build))
    assert np.array_equal(mk, m), "reference kd-tree matcher disagrees with the GPU matcher"
    kd.sort()
    stages = {"match_kdtree_ms": kd[2][0] * 1e3, "match_kdtree_build_ms": kd[2][1] * 1e3,
              "match_kdtree_note": "the reference's own matcher (PCA kd-tree, leaf 10, bestMatchFull) restated in "
                                   "oracle/vo_kdtree.c; same pairs as the GPU matcher",
              "triangulate_ms": t_tri * 1e3, "join_linear_ms": t_join * 1e3,
              "join_quadratic_ms_extrapolated": t_jq * 1e3 * len(m) / nj_s,
              "join_quadratic_sample": f"the reference's literal form (vo_complete.cpp:52-66: for each image pair scan the world "
                                       f"pairs from the start) on the first {nj_s} of {len(m)} image pairs x all "
                                       f"{len(fp['model_pairs'])} world pairs, scaled; result equal to the O(C) form: {jq_equal}",
              "match_bruteforce_ms_extrapolated": t_match * 1e3 * len(fp["cur_app"]) / nq_s,
              "match_sample": f"{nq_s} queries x {len(fp['ref_app'])} points, scaled to {len(fp['cur_app'])} queries",
              "repetitions": "median of 5 after one warm-up, every stage"}
    return {"value": args.iters / t_med, "unit": "iter/s", "cores": 1, "kind": "port", "repetitions": reps,
            "all_cores": all_cores, "march_native": march_native, "other_stages": stages,
            "sample": f"median of {reps} x {args.iters} rounds of the C float32 restatement (oracle/, gcc -O3 -ffp-contract=off) "
                      f"on the same {len(corr)}-correspondence pair; the reference itself needs Eigen3 (absent)",
            "pose_diff_gpu_vs_cpu": float(np.abs(gpu_T - r["T"]).max()), "_pose": np.asarray(r["T"], np.float32),
            "host": _cpu_model(), "host_cores_available": os.cpu_count()}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
