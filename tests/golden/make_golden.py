"""Freezes seeded inputs and the CPU oracle's outputs into small .npz fixtures.

The reference ships no golden vectors for this path (its tests draw from
std::random_device and assert nothing) and cannot be run here (needs Eigen3), so
these fixtures pin the ORACLE, not the reference: they detect drift of the
restatement and give the GPU tests byte-stable inputs/expected outputs.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import __graft_entry__ as g  # noqa: E402
from oracle.oracle import Camera, Oracle  # noqa: E402

vo = g.load_package()
o32, o64 = Oracle(32), Oracle(64)


def frame_fixture(n, seed, n_iters=10):
    fp = vo.synth.frame_pair(n, seed=seed, drop=0.1, distractors=max(4, n // 20), model_drop=0.1)
    out = {k: v for k, v in fp.items() if isinstance(v, np.ndarray)}
    out["cam_ints"] = np.array([fp["rows"], fp["cols"], fp["z_near"], fp["z_far"]], dtype=np.int32)
    m = o32.match(fp["ref_app"], fp["cur_app"])
    j = o32.join(m, fp["model_pairs"])
    out["exp_match"] = m
    out["exp_join"] = j
    cam = Camera(fp["rows"], fp["cols"], fp["z_near"], fp["z_far"], fp["K"], np.eye(4))
    for tag, thr, keep in (("a", 10000.0, False), ("b", 60.0, False), ("c", 60.0, True)):
        for bits, o in ((32, o32), (64, o64)):
            r = o.picp_solve(cam, fp["model"], fp["cur_pts"], j, n_iters, kernel_threshold=thr, keep_outliers=keep)
            out[f"picp_{tag}_H{bits}"] = r["H"]
            out[f"picp_{tag}_b{bits}"] = r["b"]
            out[f"picp_{tag}_stats{bits}"] = r["stats"]
            out[f"picp_{tag}_T{bits}"] = r["T_trace"]
    T = out["picp_a_T32"][-1]
    xyz, pairs, app = o32.triangulate(fp["K"], T, m, fp["ref_pts"], fp["cur_pts"], fp["cur_app"])
    out["exp_tri_xyz"], out["exp_tri_pairs"], out["exp_tri_app"] = xyz, pairs, app
    out["exp_transform"] = o32.transform_points(T, fp["model"])
    return out


def picp_test_fixture(seed, n_iters=20):
    s = vo.synth.picp_test_scene(seed=seed)
    out = {k: v for k, v in s.items() if isinstance(v, np.ndarray)}
    out["cam_ints"] = np.array([s["rows"], s["cols"], s["z_near"], s["z_far"]], dtype=np.int32)
    cam = Camera(s["rows"], s["cols"], s["z_near"], s["z_far"], s["K"], np.eye(4))
    uv_keep, n_in = o32.project_points(Camera(s["rows"], s["cols"], s["z_near"], s["z_far"], s["K"], s["X_gt"]),
                                       s["world"], keep_indices=True)
    uv_compact, _ = o32.project_points(Camera(s["rows"], s["cols"], s["z_near"], s["z_far"], s["K"], s["X_gt"]),
                                       s["world"], keep_indices=False)
    out["exp_proj_keep"], out["exp_proj_compact"], out["exp_proj_inside"] = uv_keep, uv_compact, np.int32(n_in)
    r = o32.picp_solve(cam, s["world"], s["cur_pts"], s["corr"], n_iters, kernel_threshold=10000.0)
    out["picp_T32"], out["picp_stats32"], out["picp_H32"], out["picp_b32"] = r["T_trace"], r["stats"], r["H"], r["b"]
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "frame64.npz"), **frame_fixture(64, 64001))
    np.savez_compressed(os.path.join(HERE, "frame1000.npz"), **frame_fixture(1000, 1000001))
    # seed 1000: every correspondence starts as an outlier (0 inliers, H = I, pose never moves);
    # seed 1009: 36 correspondences, converges to X_gt
    np.savez_compressed(os.path.join(HERE, "picp_test1000.npz"), **picp_test_fixture(1000))
    np.savez_compressed(os.path.join(HERE, "picp_test1009.npz"), **picp_test_fixture(1009, n_iters=100))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
