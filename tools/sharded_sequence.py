#!/usr/bin/env python3
"""SURVEY 8(e), second row: a real sequence is serial in its pose chain, but its matcher stage depends on
the appearances alone.  Every rank matches its contiguous block of (t-1, t) frame pairs on its own GPU,
the variable-length pair lists are exchanged with two RCCL all-gathers (dist.gather_ragged), and rank 0
runs the chain (SequencePipeline) on the precomputed matches.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
      tools/sharded_sequence.py [--data DIR | --frames F --points N] [--iters K]

--data DIR reads a dataset in the reference's format (parsed here with a few lines of numpy; nothing under
oracle/ is used).  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def read_dataset(d):
    """camera.dat / meas-*.dat of the reference's format (files_utils.cpp:29-131) -> the dict SequencePipeline takes"""
    K = np.zeros((3, 3), np.float32); H = np.eye(4, dtype=np.float32); ints = {}
    lines = open(os.path.join(d, "camera.dat")).read().splitlines()
    i = 0
    while i < len(lines):
        w = lines[i].split()
        if w and w[0] == "camera":
            K[:] = [[float(x) for x in lines[i + 1 + r].split()] for r in range(3)]; i += 4; continue
        if w and w[0] == "cam_transform:":
            H[:] = [[float(x) for x in lines[i + 1 + r].split()] for r in range(4)]; i += 5; continue
        if w and w[0] in ("z_near:", "z_far:", "width:", "height:"):
            ints[w[0][:-1]] = int(w[1])
        i += 1
    frames = []
    for f in sorted(x for x in os.listdir(d) if x.startswith("meas-") and x.endswith(".dat")):
        rows = [ln.split() for ln in open(os.path.join(d, f)).read().splitlines()[3:] if ln.strip()]
        a = np.array([[float(x) for x in r[3:15]] for r in rows], np.float32).reshape(-1, 12)
        frames.append(dict(ids=np.array([int(r[2]) for r in rows], np.int64), pts=a[:, :2].copy(), app=a[:, 2:].copy()))
    gt = np.array([[float(x) for x in ln.split()[4:7]] for ln in open(os.path.join(d, "trajectory.dat")) if ln.strip()])
    return dict(K=K, H=H, rows=ints["height"], cols=ints["width"], z_near=ints["z_near"], z_far=ints["z_far"], frames=frames,
                gt=gt[: len(frames)])


def metrics(vo, seq, traj):
    """the trajectory measures of evaluate.cpp:18-60 (README.md:33-50) against trajectory.dat / the generator:
    rotation part of rel_est^-1 rel_gt, median translation-norm ratio, position RMSE after rescaling"""
    H = seq["H"].astype(np.float64)
    Hi = np.linalg.inv(H)
    gt = [vo.synth.planar_pose(*g) for g in seq["gt"]]
    est = [np.eye(4)]
    for X in traj[1:]:                                   # save_trajectory (files_utils.cpp:136-153): W <- W C X^-1 C^-1
        est.append(est[-1] @ H @ np.linalg.inv(X.astype(np.float64)) @ Hi)
    e_th, ratio = [], []
    for i in range(1, len(gt)):
        Xr, Xg = np.linalg.inv(est[i - 1]) @ est[i], np.linalg.inv(gt[i - 1]) @ gt[i]
        e_th.append(float(np.trace(np.eye(3) - Xr[:3, :3].T @ Xg[:3, :3])))
        ng = np.linalg.norm(Xg[:3, 3])
        if ng > 0:
            ratio.append(float(np.linalg.norm(Xr[:3, 3]) / ng))
    r = float(np.median(ratio))
    rmse = float(np.sqrt(np.mean([np.linalg.norm(g[:3, 3] - e[:3, 3] / r) ** 2 for g, e in zip(gt, est)])))
    return {"mean_orientation_error": float(np.mean(e_th)), "inverse_median_ratio": 1.0 / r, "rmse_position": rmse}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--points", type=int, default=5000)
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()
    import torch
    vo = graft.load_package()
    from importlib import import_module
    vdist = import_module("visual_odometry_amd.dist")
    rank, local_rank, world = vdist.env_rank_world()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = vdist.init("nccl", local_rank)
    stream = torch.cuda.Stream(device=dev)
    ctx = vo.Context(local_rank, stream.cuda_stream)
    seq = read_dataset(args.data) if args.data else vo.synth.sequence(seed=3000, n_frames=args.frames, n_visible=args.points)
    fr = seq["frames"]
    F = len(fr)
    with torch.cuda.stream(stream):
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        lo, hi = vdist.shard_range(F - 1, rank, world)           # item k = the pair (k, k+1)
        # this rank's block of (k, k+1) pairs in ONE call, whatever the frames' sizes (vo_match_appearances_batch_dev)
        mine = [torch.from_numpy(m).to(dev) for m in vo.match_batch_ragged(ctx, [fr[k]["app"] for k in range(lo, hi)],
                                                                          [fr[k + 1]["app"] for k in range(lo, hi)])]
        every = vdist.gather_ragged(mine, device=dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = None
        if rank == 0:
            matches = [m.cpu().numpy() for m in every]
            sp = vo.SequencePipeline(ctx, seq, n_iters=args.iters, matches=matches)
            sp.run(); ctx.synchronize()
            t2 = time.perf_counter()
            traj = sp.trajectory(); counts = sp.counts(); sp.close()
            own = vo.SequencePipeline(ctx, seq, n_iters=args.iters)   # the same chain matching for itself
            own.run(); ctx.synchronize()
            same = bool(np.array_equal(own.trajectory(), traj) and np.array_equal(own.counts(), counts))
            own.close()
            out = {"frames": F, "ranks": world, "pairs_per_rank": hi - lo, "match_and_gather_ms": (t1 - t0) * 1e3,
                   "chain_ms": (t2 - t1) * 1e3, "matches_total": int(counts[1:, 0].sum()),
                   "identical_to_single_gpu_chain": same,
                   "last_pose_t": [float(x) for x in traj[-1][:3, 3]], "evaluation": metrics(vo, seq, traj)}
        dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
