"""Seeded synthetic inputs for the projective-ICP hot path.

The reference's generators (utils.cpp:8-34) draw from std::random_device, so
nothing it produces is reproducible; these keep its distributions and camera
(picp_solver_test.cpp:42-56) but are seeded (numpy PCG64), and add the
frustum-filling frame pair of SURVEY 8(d) config 2 in which every
correspondence is valid in both views.

This module only *generates data* (plain numpy, float32 outputs); it contains
no part of the checked computation.
"""
from __future__ import annotations

import numpy as np

K_REF = np.array([[180.0, 0.0, 320.0], [0.0, 180.0, 240.0], [0.0, 0.0, 1.0]], dtype=np.float32)
ROWS, COLS = 480, 640


def rodrigues(axis, angle):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * Kx + (1 - np.cos(angle)) * (Kx @ Kx)


def random_isometry(rng, max_angle=1.0, max_t=1.0):
    """generate_isometry3f (utils.cpp:8-20): axis~normalise(U(-1,1)^3),
    angle~U(-max_angle,max_angle), t~U(-max_t,max_t)^3."""
    axis = rng.uniform(-1, 1, 3)
    angle = rng.uniform(-max_angle, max_angle)
    T = np.eye(4)
    T[:3, :3] = rodrigues(axis, angle)
    T[:3, 3] = rng.uniform(-max_t, max_t, 3)
    return T.astype(np.float32)


def random_points3d(rng, n):
    """generate_points3d (utils.cpp:22-34): x,y~U(-10,10), z~U(-10,10)*0.1+1."""
    p = np.empty((n, 3), dtype=np.float32)
    p[:, 0] = rng.uniform(-10, 10, n)
    p[:, 1] = rng.uniform(-10, 10, n)
    p[:, 2] = rng.uniform(-10, 10, n).astype(np.float32) * np.float32(0.1) + np.float32(1.0)
    return p


def project_np(K, T, pts, rows, cols, z_near, z_far):
    """float64 pinhole projection used only to build inputs/margins."""
    K = np.asarray(K, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    pc = pts.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    ph = pc @ K.T
    with np.errstate(divide="ignore", invalid="ignore"):
        uv = ph[:, :2] / ph[:, 2:3]
    ok = (pc[:, 2] <= z_far) & (pc[:, 2] >= z_near)
    ok &= (uv[:, 0] >= 0) & (uv[:, 0] <= cols - 1) & (uv[:, 1] >= 0) & (uv[:, 1] <= rows - 1)
    return uv, pc, ok


def picp_test_scene(seed=1000, n=1000, f=180.0):
    """Config 1: the picp_solver_test.cpp scenario (2 poses, n world points,
    measurements by projection with keep_indices, correspondences (i,i)
    where both views are valid).  Measurements are produced in float64 and
    rounded; the (-1,-1) convention for invalid points is kept."""
    rng = np.random.default_rng(seed)
    X_gt = random_isometry(rng)
    world = random_points3d(rng, n)
    K = K_REF.copy()
    K[0, 0] = K[1, 1] = f
    uv0, _, ok0 = project_np(K, np.eye(4), world, ROWS, COLS, 0, 10)
    uv1, _, ok1 = project_np(K, X_gt, world, ROWS, COLS, 0, 10)
    ref = np.where(ok0[:, None], uv0, -1.0).astype(np.float32)
    cur = np.where(ok1[:, None], uv1, -1.0).astype(np.float32)
    idx = np.nonzero((ref[:, 0] >= 0) & (cur[:, 0] >= 0))[0].astype(np.int32)
    corr = np.stack([idx, idx], axis=1)
    return dict(K=K, rows=ROWS, cols=COLS, z_near=0, z_far=10, X_gt=X_gt, world=world,
                ref_pts=ref, cur_pts=cur, corr=corr)


def frame_pair(n, seed=2000, noise_px=0.5, drop=0.0, distractors=0, model_drop=0.0,
               max_angle=0.05, max_t=0.1, margin_px=2.0, z_lo=1.0, z_hi=9.0):
    """Config 2: a frustum-filling frame pair with n common landmarks.

    Landmarks are drawn in the CURRENT camera frame so that they fill its
    image, mapped to the REFERENCE camera frame by a small motion X_gt
    (p_cur = X_gt p_ref), rejection-sampled until n of them project inside
    both images with >= margin_px of slack and 0.25 of depth slack.

      ref_pts/ref_app : reference image (N1 points)   -- pixel + 10-D appearance
      cur_pts/cur_app : current image  (N2 points), a fixed random permutation
      model           : 3-D points in the reference frame ("triangulated")
      model_pairs     : (ref_idx, model_idx), what triangulation would emit
      gt_matches      : (ref_idx, cur_idx) ground truth, in cur (query) order

    drop: each frame independently loses this fraction of the landmarks;
    distractors: unmatched points added to each frame (different counts, so
    |a1| != |a2| either way round depending on the seed); model_drop: fraction
    of reference points without a model point (join "no partner")."""
    rng = np.random.default_rng(seed)
    K = K_REF.copy()
    X_gt = random_isometry(rng, max_angle, max_t)
    iX = np.linalg.inv(X_gt.astype(np.float64))
    pts_ref = np.empty((0, 3))
    while len(pts_ref) < n:
        m = int((n - len(pts_ref)) * 1.5) + 64
        z = rng.uniform(z_lo, z_hi, m)
        x = rng.uniform(-0.95, 0.95, m) * z * (319.5 / 180.0)
        y = rng.uniform(-0.95, 0.95, m) * z * (239.5 / 180.0)
        pc = np.stack([x, y, z], axis=1)
        pr = (pc @ iX[:3, :3].T + iX[:3, 3]).astype(np.float32)
        ok = np.ones(m, dtype=bool)
        for T in (np.eye(4), X_gt):
            uv, pcam, _ = project_np(K, T, pr, ROWS, COLS, 0, 10)
            ok &= (uv[:, 0] >= margin_px) & (uv[:, 0] <= COLS - 1 - margin_px)
            ok &= (uv[:, 1] >= margin_px) & (uv[:, 1] <= ROWS - 1 - margin_px)
            ok &= (pcam[:, 2] >= 0.25) & (pcam[:, 2] <= 9.75)
        pts_ref = np.concatenate([pts_ref, pr[ok]], axis=0)
    pts_ref = pts_ref[:n].astype(np.float32)
    app = rng.uniform(-1, 1, (n, 10)).astype(np.float32)

    uv_ref, _, _ = project_np(K, np.eye(4), pts_ref, ROWS, COLS, 0, 10)
    uv_cur, _, _ = project_np(K, X_gt, pts_ref, ROWS, COLS, 0, 10)
    uv_ref = uv_ref + rng.normal(0, noise_px, uv_ref.shape)
    uv_cur = uv_cur + rng.normal(0, noise_px, uv_cur.shape)

    keep_ref = rng.uniform(size=n) >= drop
    keep_cur = rng.uniform(size=n) >= drop
    ids_ref = np.nonzero(keep_ref)[0]
    ids_cur = rng.permutation(np.nonzero(keep_cur)[0])
    d_ref = int(distractors)
    d_cur = int(distractors * 2 + (seed % 3)) if distractors else 0

    def with_distractors(uv, ap, d):
        if d == 0:
            return uv.astype(np.float32), ap.astype(np.float32)
        duv = np.stack([rng.uniform(0, COLS - 1, d), rng.uniform(0, ROWS - 1, d)], axis=1)
        dap = rng.uniform(-1, 1, (d, 10))
        return (np.concatenate([uv, duv]).astype(np.float32),
                np.concatenate([ap, dap]).astype(np.float32))

    ref_pts, ref_app = with_distractors(uv_ref[ids_ref], app[ids_ref], d_ref)
    cur_pts, cur_app = with_distractors(uv_cur[ids_cur], app[ids_cur], d_cur)

    # model points: one per kept reference point, minus model_drop, in a shuffled order
    has_model = rng.uniform(size=len(ids_ref)) >= model_drop
    ref_with_model = np.nonzero(has_model)[0]
    order = rng.permutation(len(ref_with_model)) if (drop or model_drop or distractors) else np.arange(len(ref_with_model))
    model = pts_ref[ids_ref[ref_with_model[order]]]
    model_pairs = np.stack([ref_with_model[order], np.arange(len(order))], axis=1).astype(np.int32)

    pos_in_ref = -np.ones(n, dtype=np.int64)
    pos_in_ref[ids_ref] = np.arange(len(ids_ref))
    gt = [(pos_in_ref[l], j) for j, l in enumerate(ids_cur) if pos_in_ref[l] >= 0]
    gt_matches = np.array(gt, dtype=np.int32).reshape(-1, 2)
    return dict(K=K, rows=ROWS, cols=COLS, z_near=0, z_far=10, X_gt=X_gt,
                ref_pts=ref_pts, ref_app=ref_app, cur_pts=cur_pts, cur_app=cur_app,
                model=np.ascontiguousarray(model, dtype=np.float32), model_pairs=model_pairs,
                gt_matches=gt_matches)


# ---- config 3: a synthetic SEQUENCE in the reference's dataset conventions ----------------
CAM_IN_ROBOT = np.array([[0, 0, 1, 0.2], [-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)


def planar_pose(x, y, th):
    T = np.eye(4)
    c, s = np.cos(th), np.sin(th)
    T[:2, :2] = [[c, -s], [s, c]]
    T[:2, 3] = [x, y]
    return T


def sequence(seed=3000, n_frames=200, n_visible=2000, step=0.2, yaw_amp=0.015, noise_px=0.0, z_vis=5.0,
             z_lo=-1.5, z_hi=2.5):
    """a planar robot advancing `step` per frame with a slowly varying yaw
    (the shape of example_data's trajectory.dat), a camera mounted as in camera.dat
    (cam_transform: optical axis along the robot's x), and landmarks spread uniformly in a slab
    around the path at a density that puts about n_visible of them in view per frame.  Every
    landmark carries a 10-D appearance that is copied exactly into each measurement (as in the
    real data), each frame lists its measurements in its own random order.

    The reference's Camera gate z <= z_far (camera.h:28) is applied by the solver to model
    points in the ESTIMATED scale, which the epipolar step fixes at |t_0| = singular value of E
    (un-normalised, epipolar_utils.cpp:163-164; about 0.3-0.4 with this camera), i.e. |t_0|/step
    times the true one; the camera file gets z_far = ceil(z_vis/step)+1, which leaves headroom
    for any |t_0| <= 1, while visibility is generated with the true cut-off z_vis.

    Returns a dict: K, rows, cols, z_near, z_far, H (camera in robot), gt (n_frames,3: x,y,theta),
    world_xyz (M,3), world_app (M,10), frames = [dict(ids, pts (n,2), app (n,10))]."""
    rng = np.random.default_rng(seed)
    F = int(n_frames)
    th = np.zeros(F); xy = np.zeros((F, 2))
    phase = rng.uniform(0, 2 * np.pi)
    for t in range(1, F):
        th[t] = th[t - 1] + yaw_amp * np.sin(2 * np.pi * t / 60.0 + phase)
        xy[t] = xy[t - 1] + step * np.array([np.cos(th[t - 1]), np.sin(th[t - 1])])
    gt = np.concatenate([xy, th[:, None]], axis=1)
    K = K_REF.astype(np.float64)
    reach = float(np.hypot(z_vis, z_vis * (COLS / 2) / K[0, 0])) + 0.5       # farthest visible point from the camera
    lo = np.array([xy[:, 0].min() - reach, xy[:, 1].min() - reach, z_lo])
    hi = np.array([xy[:, 0].max() + reach, xy[:, 1].max() + reach, z_hi])
    T_cw = [np.linalg.inv(planar_pose(*gt[t]) @ CAM_IN_ROBOT) for t in range(F)]

    def visible(P, t):
        pc = P @ T_cw[t][:3, :3].T + T_cw[t][:3, 3]
        z = pc[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = K[0, 0] * pc[:, 0] / z + K[0, 2]
            v = K[1, 1] * pc[:, 1] / z + K[1, 2]
        ok = (z > 1e-3) & (z <= z_vis) & (u >= 0) & (u <= COLS - 1) & (v >= 0) & (v <= ROWS - 1)
        return ok, u, v

    pilot = rng.uniform(lo, hi, (100000, 3))
    seen = np.mean([visible(pilot, t)[0].sum() for t in range(0, F, max(F // 8, 1))])
    M = max(int(round(100000 * n_visible / max(seen, 1.0))), 16)
    world = rng.uniform(lo, hi, (M, 3))
    order = np.argsort(world[:, 0], kind="stable")
    world = np.ascontiguousarray(world[order]).astype(np.float32)               # float32 is what world.dat stores
    w64 = world.astype(np.float64)
    app = rng.uniform(-1, 1, (M, 10)).astype(np.float32)
    frames = []
    for t in range(F):
        a, b = np.searchsorted(w64[:, 0], [xy[t, 0] - reach, xy[t, 0] + reach])
        ok, u, v = visible(w64[a:b], t)
        ids = a + np.nonzero(ok)[0]
        uv = np.stack([u[ok], v[ok]], axis=1)
        if noise_px:
            uv = uv + rng.normal(0, noise_px, uv.shape)
        p = rng.permutation(len(ids))
        frames.append(dict(ids=ids[p].astype(np.int64), pts=uv[p].astype(np.float32), app=app[ids[p]]))
    return dict(K=K_REF.copy(), rows=ROWS, cols=COLS, z_near=0, z_far=int(np.ceil(z_vis / step)) + 1,
                H=CAM_IN_ROBOT.astype(np.float32), gt=gt, world_xyz=world, world_app=app, frames=frames, step=step)


def sequence_gt_relative(seq):
    """X_gt[t] (t >= 1): pose of camera t-1 in the frame of camera t, true scale (what the solver
    estimates up to the global monocular scale)."""
    H = seq["H"].astype(np.float64)
    Twc = [planar_pose(*g) @ H for g in seq["gt"]]
    return [np.linalg.inv(Twc[t]) @ Twc[t - 1] for t in range(1, len(Twc))]


def write_sequence(seq, out_dir):
    """Write `seq` in the reference's dataset format (files_utils.cpp:29-131): camera.dat,
    trajectory.dat (k, odometry x y theta, ground truth x y theta), world.dat (id xyz app) and
    meas-XXXXX.dat (3 header lines, then `point <k> <id> <col> <row> <10 appearance>`).  Floats
    are printed with 9 significant digits, so every reader recovers the same float32."""
    import os
    os.makedirs(out_dir, exist_ok=True)
    g = lambda a: " ".join("%.9g" % float(x) for x in a)
    K, H = seq["K"], seq["H"]
    with open(os.path.join(out_dir, "camera.dat"), "w") as f:
        f.write("camera matrix:\n" + "".join(g(K[r]) + "\n" for r in range(3)))
        f.write("cam_transform:\n" + "".join(g(H[r]) + "\n" for r in range(4)))
        f.write("z_near: %d\nz_far:  %d\nwidth:  %d\nheight: %d\n" % (seq["z_near"], seq["z_far"], seq["cols"], seq["rows"]))
    with open(os.path.join(out_dir, "trajectory.dat"), "w") as f:
        for k, p in enumerate(seq["gt"]):
            f.write("%d %s %s\n" % (k, g(p), g(p)))
    with open(os.path.join(out_dir, "world.dat"), "w") as f:
        for i, (p, a) in enumerate(zip(seq["world_xyz"], seq["world_app"])):
            f.write("%d %s %s\n" % (i, g(p), g(a)))
    for t, fr in enumerate(seq["frames"]):
        with open(os.path.join(out_dir, "meas-%05d.dat" % t), "w") as f:
            f.write("seq: %d\ngt_pose: %s\nodom_pose: %s\n" % (t, g(seq["gt"][t]), g(seq["gt"][t])))
            for k, (i, p, a) in enumerate(zip(fr["ids"], fr["pts"], fr["app"])):
                f.write("point %d %d %s %s\n" % (k, int(i), g(p), g(a)))
