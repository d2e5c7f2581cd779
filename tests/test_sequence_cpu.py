"""SURVEY 8(d) config 3, CPU side: the synthetic sequence generator, its dataset writer (read back by
the oracle's readers) and the oracle's vo_complete restatement on it (no GPU)."""
import numpy as np

from oracle import vo_pipeline as P


def test_sequence_generator_is_deterministic_and_consistent(vo):
    a = vo.synth.sequence(seed=3000, n_frames=8, n_visible=200)
    b = vo.synth.sequence(seed=3000, n_frames=8, n_visible=200)
    c = vo.synth.sequence(seed=3001, n_frames=8, n_visible=200)
    assert len(a["frames"]) == 8 and a["z_far"] == 26
    for fa, fb in zip(a["frames"], b["frames"]):
        assert np.array_equal(fa["ids"], fb["ids"]) and np.array_equal(fa["pts"], fb["pts"])
    assert not np.array_equal(a["world_xyz"][:50], c["world_xyz"][:50])
    n = [len(f["ids"]) for f in a["frames"]]
    assert 120 < min(n) and max(n) < 300, n
    for f in a["frames"]:
        assert len(set(f["ids"].tolist())) == len(f["ids"])                     # a landmark is seen once per frame
        assert np.array_equal(f["app"], a["world_app"][f["ids"]])               # appearance copied bit for bit
        assert (f["pts"][:, 0] >= 0).all() and (f["pts"][:, 0] <= a["cols"] - 1).all()
        assert (f["pts"][:, 1] >= 0).all() and (f["pts"][:, 1] <= a["rows"] - 1).all()
    # measurements are the projections of the landmarks through the ground-truth camera
    T = np.linalg.inv(vo.synth.planar_pose(*a["gt"][3]) @ a["H"].astype(np.float64))
    f = a["frames"][3]
    pc = a["world_xyz"][f["ids"]].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    uv = (pc @ a["K"].astype(np.float64).T)
    assert np.abs(uv[:, :2] / uv[:, 2:3] - f["pts"]).max() < 1e-3
    # consecutive relative poses: |t| = step, rotation about the camera's y axis only
    for X in vo.synth.sequence_gt_relative(a):
        assert abs(np.linalg.norm(X[:3, 3]) - a["step"]) < 1e-2 and abs(X[1, 1] - 1) < 1e-12


def test_written_dataset_round_trips_and_oracle_tracks_it(vo, o32, tmp_path):
    seq = vo.synth.sequence(seed=3000, n_frames=12, n_visible=250)
    d = str(tmp_path)
    vo.synth.write_sequence(seq, d)
    K, H, ints = P.read_camera(d + "/camera.dat")
    assert np.array_equal(K, seq["K"]) and np.array_equal(H, seq["H"])
    assert ints == dict(z_near=0, z_far=26, width=640, height=480)
    for t in (0, 5, 11):
        pts, app, ids = P.read_meas(d + "/meas-%05d.dat" % t)
        f = seq["frames"][t]
        assert np.array_equal(pts, f["pts"]) and np.array_equal(app, f["app"]) and np.array_equal(ids, f["ids"])
    w, wa = P.read_world(d + "/world.dat")
    assert np.array_equal(w, seq["world_xyz"]) and np.array_equal(wa, seq["world_app"])
    gt = P.read_gt(d + "/trajectory.dat")
    assert np.allclose(gt[7], vo.synth.planar_pose(*seq["gt"][7]), atol=1e-8)
    res = P.run_vo_complete(d, rounds=100, o=o32)
    # structural KAT: appearance matches = overlap of the ground-truth ids of consecutive frames
    for t in range(2, 12):
        common = len(set(seq["frames"][t - 1]["ids"].tolist()) & set(seq["frames"][t]["ids"].tolist()))
        assert res["stats"][t - 2][0] == common
        assert res["stats"][t - 2][2] == res["stats"][t - 2][1]                 # noise-free: every joined pair is an inlier
    m = P.evaluate(d, res)
    assert abs(m["mean_orientation_error"]) < 1e-5
    assert m["rmse_position"] < 0.02 and m["rmse_map"] < 0.15, m
