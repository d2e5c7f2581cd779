#!/usr/bin/env python3
"""bench.py -- PICP iterations/sec at 50 000 correspondences on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

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

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_CORR_ITER = 20       # world xyz 12 B + measurement uv 8 B


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch-pairs", type=int, default=200, help="problems of the batched-solver leg (0: skip)")
    ap.add_argument("--frame-steps", type=int, default=30, help="frames of the whole-frame leg (0: skip)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU-baseline leg (0: skip)")
    ap.add_argument("--no-extras", action="store_true", help="headline measurement only")
    ap.add_argument("--legs", default=None,
                    help="extra legs to run (comma list of frame,batched,sequence,cpu); default: all four on one "
                         "GPU, only `frame` (the sharded config-4 leg) when several ranks run")
    ap.add_argument("--seq-frames", type=int, default=200, help="frames of the sequence leg (config 3)")
    ap.add_argument("--seq-points", type=int, default=50000, help="landmarks in view per frame in the sequence leg")
    ap.add_argument("--seq-iters", type=int, default=100, help="PICP rounds per frame in the sequence leg (vo_complete.cpp:163)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    vo = graft.load_package()
    from importlib import import_module
    vdist = import_module("visual_odometry_amd.dist")
    rank, local_rank, world = vdist.env_rank_world()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the RCCL path runs even with one rank, so that a
    # 1-GPU box exercises exactly the code the N-GPU launch uses
    dist = vdist.init("nccl", local_rank) if (world > 1 or "RANK" in os.environ) else None

    stream = torch.cuda.Stream(device=dev)
    ctx = vo.Context(local_rank, stream.cuda_stream)
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
        "metric": "PICP iterations/sec @50k pts",
        "value": value, "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
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
                     "note": "one launch = one Gauss-Newton round over one 50k pair (1.0 MB, L2-resident): "
                             "latency-bound by the serial solve->linearize dependency, not by HBM; launch_us is "
                             "event time over the timed region / launches, i.e. it includes the kernel boundary"},
    }

    # the rank-0-only legs (batched solver sweep, serial sequence, CPU baseline) would keep the other ranks
    # spinning in the final barrier: with several ranks only the sharded leg runs unless asked for explicitly
    default_legs = "frame,batched,sequence,cpu" if world == 1 else "frame"
    legs = set() if args.no_extras else set((args.legs or default_legs).split(","))
    if "frame" in legs and args.frame_steps > 0:           # every rank: config 4 (sharded pairs + pose gather)
        with torch.cuda.stream(stream):
            cfg4 = frame_throughput(vo, torch, ctx, stream, args, dist, vdist, rank, world)
        out["batched_frames"] = cfg4
    if rank == 0 and legs:
        with torch.cuda.stream(stream):
            if args.frame_steps > 0 and "frame" in legs:
                out["frame"] = frame_leg(torch, ctx, stream, pipe, fp, args, vo)
            if args.batch_pairs > 0 and "batched" in legs:
                out["batched"] = batched_leg(torch, vo, ctx, stream, args)
            if args.seq_frames >= 3 and "sequence" in legs:
                out["sequence"] = sequence_leg(vo, ctx, args)
        if args.cpu_seconds > 0 and world == 1 and "cpu" in legs:
            out["cpu_baseline"] = cpu_leg(fp, pipe, args)
    pipe.close()
    if dist is not None:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def _pmc_traffic(kernel, grid_threads):
    """HBM bytes per launch from the committed PMC summary (profiles/r01_pmc_fetch_write_v3.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very script), for the launch
    geometry `grid_threads`.  Returns (bytes, note) or (None, reason)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_v3.json")
    try:
        k = json.load(open(path))["kernels"][kernel]
        g = k["by_grid"][str(grid_threads)]
        fetch_kb, write_kb = g["FETCH_SIZE_KB_max"], g["WRITE_SIZE_KB_max"]
    except (OSError, KeyError, ValueError):
        return None, "no committed PMC summary for this kernel and launch geometry"
    wide = k.get("wide_16B_loads", False)
    b = (2.0 * fetch_kb if wide else fetch_kb) * 1024.0 + write_kb * 1024.0
    return b, ("FETCH_SIZE x2 (gfx950 counts half of 16-B/lane coalesced reads) + WRITE_SIZE" if wide else
               "FETCH_SIZE + WRITE_SIZE as reported (4-B/lane loads: uncalibrated width)") + ", from " + os.path.basename(path)


def _chk(lib, rc):
    if rc != 0:
        raise RuntimeError(lib.vo_last_error().decode())


def frame_throughput(vo, torch, ctx, stream, args, dist=None, vdist=None, rank=0, world=1, frames=200):
    """BASELINE configs[3]: `frames` independent frame pairs per GPU resident in HBM, processed by
    vo_frames_batch_dev (every stage one batched launch, frame = a grid dimension; batched solver); with
    several ranks every rank owns its block of pairs and the job ends with ONE RCCL all-gather of the poses.
    Called by every rank; timing bracketed by barrier + synchronize, max over ranks."""
    distinct = [vo.synth.frame_pair(args.points, seed=6000 + 4 * rank + i) for i in range(4)]
    fps_in = [distinct[i % 4] for i in range(frames)]      # distinct copies in HBM; values repeat every 4 frames
    poses_t = torch.zeros((frames, 16), dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    bp = vo.BatchPipeline(ctx, fps_in, n_iters=args.iters, poses_ptr=poses_t.data_ptr())
    bp.run()
    ctx.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    reps = 3
    barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        bp.run()
        if dist is not None:
            all_poses = vdist.gather_poses(poses_t)        # (world*frames, 16), rank-major = global pair order
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt = vdist.max_over_ranks(dt, poses_t.device)
        assert all_poses.shape == (world * frames, 16)
        assert torch.equal(all_poses[rank * frames:(rank + 1) * frames], poses_t)
    ms = dt * 1e3 / reps
    P = bp.poses()
    err = max(float(np.abs(P[i] - fps_in[i]["X_gt"]).max()) for i in range(frames))
    c = bp.counts()
    assert err < 1e-3 and int(c[1].min()) == args.points, (err, c[:, :4])
    bp.close()
    return {"frames_per_gpu": frames, "n_gpus": world, "frames_total": frames * world, "ms_per_batch": ms,
            "frames_per_sec": frames * world / (ms * 1e-3), "us_per_frame_per_gpu": ms * 1e3 / frames, "pose_err_vs_gt": err,
            "note": "match + join + transform + 50 rounds + triangulate for every frame, one vo_frames_batch_dev call per "
                    "rank" + ("; + one all_gather_into_tensor of the poses per batch" if dist is not None else "")}


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
    return {"frames_per_sec": args.frame_steps / dt, "ms_per_frame": dt * 1e3 / args.frame_steps, 
            "counts": {"matches": c[0], "joined": c[1], "triangulated": c[2]}, **stages,
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
    def timed(overlap):
        sp = vo.SequencePipeline(ctx, seq, n_iters=args.seq_iters, overlap_match=overlap)
        sp.run(); ctx.synchronize()                  # warm-up pass: sizes every workspace, builds the solver graph
        t0 = time.perf_counter()
        sp.start(); ctx.synchronize()
        t1 = time.perf_counter()
        for t in range(2, sp.F):
            sp.step(t)
        ctx.synchronize()
        t2 = time.perf_counter()
        res = (sp.trajectory(), sp.counts(), t1 - t0, t2 - t1, sp.F)
        sp.close()
        return res

    traj, counts, init_s, chain_s, F = timed(False)
    traj2, counts2, _, chain2, _ = timed(True)
    assert np.array_equal(traj, traj2) and np.array_equal(counts, counts2), "overlapped matcher changed the result"
    m = _sequence_metrics(vo, seq, traj)
    n = [len(f["pts"]) for f in seq["frames"]]
    return {"frames": F, "points_per_frame": {"min": int(min(n)), "max": int(max(n))}, "landmarks": len(seq["world_xyz"]),
            "iters_per_frame": args.seq_iters, "init_ms": init_s * 1e3,
            "chain_ms": chain_s * 1e3, "frames_per_sec": (F - 2) / chain_s,
            "ms_per_frame": chain_s * 1e3 / (F - 2),
            "picp_iters_per_sec": (F - 2) * args.seq_iters / chain_s,
            "frames_per_sec_matcher_on_second_stream": (F - 2) / chain2,
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
        for P in (256, 512):
            r = _batched_run(torch, vo, ctx, stream, args, P)
            out["chip_full"].append({"pairs": P, "kernel_ms": r["kernel_ms"], "iters_per_sec": r["iters_per_sec"],
                                     "achieved_GBs": r["roofline"]["achieved"], "frac": r["roofline"]["frac"]})
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
    return {"pairs": P, "ms_per_call": ms, "pack_ms": pack_ms, "kernel_ms": kernel_ms,
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


def cpu_leg(fp, pipe, args):
    """The oracle's float32 restatement of PICPSolver::oneRound on the host, 1 thread
    (the reference has no threading), same 50k pair, same 50 rounds."""
    from oracle.oracle import Camera as OCam, Oracle
    o = Oracle(32)
    corr = pipe.fetch("join")
    cam = OCam(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4))
    o.picp_solve(cam, fp["model"], fp["cur_pts"], corr, 2, 10000.0, False, trace=False)
    runs, t_used = 0, 0.0
    while t_used < args.cpu_seconds and runs < 1000:
        t0 = time.perf_counter()
        r = o.picp_solve(cam, fp["model"], fp["cur_pts"], corr, args.iters, 10000.0, False, trace=False)
        t_used += time.perf_counter() - t0
        runs += 1
    # the strong baseline: the same loop over all the cores this job may use (per-thread partial sums)
    try:
        n_thr = len(os.sched_getaffinity(0))
    except AttributeError:
        n_thr = os.cpu_count() or 1
    n_thr = max(1, min(n_thr, 16))                 # a 1-GPU box's CPU share is 16 cores
    o.picp_solve_mt(cam, fp["model"], fp["cur_pts"], corr, args.iters, n_thr, 10000.0)      # spins the team up
    mt_runs, mt_used = 0, 0.0
    while mt_used < min(args.cpu_seconds, 5.0) and mt_runs < 1000:
        t0 = time.perf_counter()
        rm = o.picp_solve_mt(cam, fp["model"], fp["cur_pts"], corr, args.iters, n_thr, 10000.0)
        mt_used += time.perf_counter() - t0
        mt_runs += 1
    all_cores = {"value": mt_runs * args.iters / mt_used, "unit": "iter/s", "cores": rm["threads"],
                 "kind": "port, OpenMP over contiguous chunks with per-thread H/b partials (the reference itself has no threading)",
                 "pose_diff_vs_single_thread": float(np.abs(rm["T"] - r["T"]).max())}
    gpu_T = pipe.pose()
    # other stages of the frame on the host, bounded samples
    m = pipe.fetch("match")
    t0 = time.perf_counter(); o.triangulate(fp["K"], gpu_T, m, fp["ref_pts"], fp["cur_pts"]); t_tri = time.perf_counter() - t0
    t0 = time.perf_counter(); o.join(m, fp["model_pairs"], linear=True); t_join = time.perf_counter() - t0
    nq_s = min(200, len(fp["cur_app"]))
    t0 = time.perf_counter(); o.match(fp["ref_app"], fp["cur_app"][:nq_s]); t_match = (time.perf_counter() - t0) * len(fp["cur_app"]) / nq_s
    mk, t_build, t_query = o.match_kdtree(fp["ref_app"], fp["cur_app"], timing=True)
    assert np.array_equal(mk, m), "reference kd-tree matcher disagrees with the GPU matcher"
    stages = {"match_kdtree_ms": (t_build + t_query) * 1e3, "match_kdtree_build_ms": t_build * 1e3,
              "match_kdtree_note": "the reference's own matcher (PCA kd-tree, leaf 10, bestMatchFull) restated in "
                                   "oracle/vo_kdtree.c; same pairs as the GPU matcher",
              "triangulate_ms": t_tri * 1e3, "join_linear_ms": t_join * 1e3,
              "match_bruteforce_ms_extrapolated": t_match * 1e3,
              "match_sample": f"{nq_s} queries x {len(fp['ref_app'])} points, scaled to {len(fp['cur_app'])} queries"}
    return {"value": runs * args.iters / t_used, "unit": "iter/s", "cores": 1, "kind": "port", "all_cores": all_cores,
            "other_stages": stages,
            "sample": f"{runs} x {args.iters} rounds of the C float32 restatement (oracle/, gcc -O3 -ffp-contract=off) "
                      f"on the same {len(corr)}-correspondence pair; the reference itself needs Eigen3 (absent)",
            "pose_diff_gpu_vs_cpu": float(np.abs(gpu_T - r["T"]).max()),
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
