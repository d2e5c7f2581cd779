"""SURVEY 8(d) config 3 on the GPU: a synthetic sequence in the reference's dataset format, run
(1) through the device-resident chain (SequencePipeline), (2) through the C++ vo_complete
counterpart, and checked against the oracle's vo_complete restatement on the same files."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES, N_VISIBLE, ROUNDS = 24, 300, 100


@pytest.fixture(scope="module")
def seq_run(vo, o32, tmp_path_factory):
    from oracle import vo_pipeline as P
    d = str(tmp_path_factory.mktemp("seq"))
    seq = vo.synth.sequence(seed=3000, n_frames=N_FRAMES, n_visible=N_VISIBLE)
    vo.synth.write_sequence(seq, d)
    res = P.run_vo_complete(d, rounds=ROUNDS, o=o32)
    return seq, d, res, P


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def test_estimate_transform_matches_oracle(vo, o32, ctx, seq_run):
    """vo_estimate_transform (host double Jacobi + GPU cheirality vote) against the oracle's numpy-SVD
    restatement of epipolar_utils.cpp:176-213 on the first pair.  Tolerance 2e-5 abs on R and t: both
    solve the same 9x9 null-space problem in double and round once to float32."""
    seq, d, res, P = seq_run
    f0, f1 = seq["frames"][0], seq["frames"][1]
    corr = vo.compute_correspondences_images(f0["app"], f1["app"], ctx=ctx)
    assert np.array_equal(corr, o32.match(f0["app"], f1["app"]))
    X = vo.estimate_transform(seq["K"], corr, f0["pts"], f1["pts"], ctx=ctx)
    Xo = P.estimate_transform(o32, seq["K"], corr, f0["pts"], f1["pts"])
    assert _rel(X, Xo) < 2e-5, (X, Xo)
    # direction of motion: the first camera lies behind the second along its optical axis
    Xg = vo.synth.sequence_gt_relative(seq)[0]
    t, tg = X[:3, 3] / np.linalg.norm(X[:3, 3]), Xg[:3, 3] / np.linalg.norm(Xg[:3, 3])
    assert float(t @ tg) > 0.9999 and _rel(X[:3, :3], Xg[:3, :3]) < 1e-3


def test_estimate_transform_errors(vo, ctx):
    rng = np.random.default_rng(0)
    p = rng.uniform(0, 400, (20, 2)).astype(np.float32)
    with pytest.raises(vo.VoError) as e:
        vo.estimate_transform(np.eye(3), np.stack([np.arange(7)] * 2, 1), p, p, ctx=ctx)
    assert e.value.code == -1
    bad = np.stack([np.arange(9), np.arange(9)], 1); bad[3, 1] = 20
    with pytest.raises(vo.VoError) as e:
        vo.estimate_transform(np.eye(3), bad, p, p, ctx=ctx)
    assert e.value.code == -5


def test_sequence_chain_matches_oracle(vo, ctx, seq_run):
    """Device-resident chain vs the oracle pipeline, frame by frame.  Counts (matches, joined) are exact.
    Poses: the chain feeds each frame's triangulation into the next solve, so float32 reduction-order
    differences propagate; with ~250 well-spread inliers per frame they stay small -- tolerance
    5e-4 abs on R and 5e-4 * max(1,|t|) on t over all 24 frames (measured: see assertion message)."""
    seq, d, res, P = seq_run
    sp = vo.SequencePipeline(ctx, seq, n_iters=ROUNDS, keep_appearance=True)
    sp.run()
    traj, counts = sp.trajectory(), sp.counts()
    assert len(traj) == len(res["trajectory"]) == N_FRAMES
    for t in range(2, N_FRAMES):
        assert (counts[t, 0], counts[t, 1]) == res["stats"][t - 2][:2], (t, counts[t], res["stats"][t - 2])
    worst_R = max(_rel(a[:3, :3], b[:3, :3]) for a, b in zip(traj, res["trajectory"]))
    worst_t = max(_rel(a[:3, 3], b[:3, 3]) / max(1.0, float(np.linalg.norm(b[:3, 3]))) for a, b in zip(traj, res["trajectory"]))
    assert worst_R < 5e-4 and worst_t < 5e-4, (worst_R, worst_t)
    assert sp.stats()[2] == res["stats"][-1][2]                       # inliers of the last solve
    # the last cloud against the oracle's last triangulation is not kept by run_vo_complete; check its size
    xyz, pairs, app = sp.cloud(N_FRAMES - 1)
    assert len(xyz) == counts[-1, 2] > 0 and np.array_equal(pairs[:, 1], np.arange(len(xyz)))
    f = seq["frames"][-1]
    assert np.array_equal(app, f["app"][pairs[:, 0]])
    sp.close()


def test_sequence_accuracy(vo, ctx, seq_run):
    """noise-free synthetic data: the estimate must follow the generator's ground truth"""
    seq, d, res, P = seq_run
    sp = vo.SequencePipeline(ctx, seq, n_iters=ROUNDS)
    sp.run()
    traj = sp.trajectory()
    sp.close()
    Xgt = vo.synth.sequence_gt_relative(seq)
    ratio = [np.linalg.norm(traj[t][:3, 3]) / np.linalg.norm(Xgt[t - 1][:3, 3]) for t in range(1, N_FRAMES)]
    assert max(ratio) / min(ratio) < 1.02, (min(ratio), max(ratio))                    # no scale drift
    for t in range(1, N_FRAMES):
        assert _rel(traj[t][:3, :3], Xgt[t - 1][:3, :3]) < 2e-3, t


def test_cpp_vo_complete_on_synthetic_sequence(vo, seq_run, tmp_path):
    """the C++ application (facade + file I/O + map + evaluation) on the same files: its trajectory file
    against the oracle's robot trajectory, its metrics against the oracle's evaluate()."""
    seq, d, res, P = seq_run
    exe, ev = os.path.join(ROOT, "apps", "bin", "vo_complete"), os.path.join(ROOT, "apps", "bin", "evaluate")
    if not (os.path.exists(exe) and os.path.exists(ev)):
        pytest.skip("apps not built")
    out = str(tmp_path) + "/"
    subprocess.run([exe, d, out, str(ROUNDS)], check=True, stdout=subprocess.DEVNULL, timeout=600)
    r = subprocess.run([ev, d, out], check=True, capture_output=True, text=True, timeout=600)
    est = np.loadtxt(os.path.join(out, "trajectory_est_complete.txt"))
    ref = np.array([W[:3, 3] for W in P.robot_trajectory(res["trajectory"], res["H"])])
    assert est.shape == ref.shape == (N_FRAMES, 3)
    assert _rel(est, ref) < 2e-3, _rel(est, ref)
    m = P.evaluate(d, res)
    val = {k: float(v) for k, v in re.findall(r"^(.*?):\s*([-0-9.e+]+)", r.stdout, flags=re.M)}
    assert abs(val["ratio used for map correction"] - m["median_ratio_inv"]) < 2e-3 * m["median_ratio_inv"], (val, m)
    assert abs(val["RMSE position"] - m["rmse_position"]) < 0.05 * m["rmse_position"] + 1e-3, (val, m)
    assert abs(val["RMSE map"] - m["rmse_map"]) < 0.05 * m["rmse_map"] + 1e-3, (val, m)


def test_upfront_sharded_matching_over_rccl(vo, tmp_path):
    """tools/sharded_sequence.py under torch.distributed.run (one rank on this box, so the RCCL path of
    dist.gather_ragged runs): matches computed up front + exchanged give the identical chain."""
    import json
    import socket
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tools", "sharded_sequence.py"), "--frames", "10", "--points", "300"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["frames"] == 10 and out["ranks"] == 1 and out["identical_to_single_gpu_chain"] is True
    assert out["matches_total"] > 9 * 150
    assert abs(out["evaluation"]["mean_orientation_error"]) < 1e-5
    # BASELINE configs[4] in its one-rank form: the reference's example sequence, matcher stage up front over RCCL,
    # chain on rank 0; the README's scale figure (README.md:74-79; the chain's tail is chaotic)
    data = os.path.join(ROOT, "tests", "golden", "example_data", "data")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tools", "sharded_sequence.py"), "--data", data],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["frames"] == 121 and out["identical_to_single_gpu_chain"] is True
    assert abs(out["evaluation"]["inverse_median_ratio"] - 0.47337) < 0.015 * 0.47337, out["evaluation"]
    assert out["evaluation"]["rmse_position"] < 0.145332 * 1.5, out["evaluation"]


def test_overlapped_matcher_gives_identical_chain(vo, ctx, seq_run):
    """overlap_match=True runs every matcher one frame ahead on a second context (own stream), ordered by
    vo_event_* against the chain: counts and poses must be bit-identical to the single-stream chain, also
    when the same object is run twice (buffers and events are reused)."""
    seq, d, res, P = seq_run
    a = vo.SequencePipeline(ctx, seq, n_iters=ROUNDS)
    a.run()
    ta, ca = a.trajectory(), a.counts()
    a.close()
    b = vo.SequencePipeline(ctx, seq, n_iters=ROUNDS, overlap_match=True)
    for _ in range(2):
        b.run()
        assert np.array_equal(b.trajectory(), ta) and np.array_equal(b.counts(), ca)
    b.close()
    # events order work across contexts without blocking the host; misuse is reported
    e = vo.Event(ctx)
    e.record(ctx); e.wait(ctx); ctx.synchronize(); e.close()
    assert ctx.lib.vo_event_record(None, ctx.h) == -1


def test_sequence_matched_up_front_gives_identical_chain(vo, ctx, seq_run):
    """prematch=True matches every consecutive pair of the sequence in ONE ragged batched call at start()
    (vo_match_appearances_batch_dev; the frames hold different numbers of points) and the chain reads pairs
    and counts where that call left them: counts, poses and clouds bit-identical to the frame-by-frame chain, twice."""
    seq, d, res, P = seq_run
    a = vo.SequencePipeline(ctx, seq, n_iters=ROUNDS)
    a.run()
    ta, ca, cloud_a = a.trajectory(), a.counts(), a.cloud(a.F - 1)
    a.close()
    b = vo.SequencePipeline(ctx, seq, n_iters=ROUNDS, prematch=True)
    for _ in range(2):
        b.run()
        assert np.array_equal(b.counts(), ca) and np.array_equal(b.trajectory(), ta)
        cloud_b = b.cloud(b.F - 1)
        assert np.array_equal(cloud_b[0], cloud_a[0]) and np.array_equal(cloud_b[1], cloud_a[1])
    b.close()


def test_cpp_resident_sequence_equals_frame_by_frame(vo, seq_run, tmp_path):
    """apps/vo_complete --resident (vo::DeviceSequence: whole chain on the GPU, map built afterwards) writes the
    same trajectory and map files as the frame-by-frame facade loop, and the same trajectory as SequencePipeline."""
    seq, d, res, P = seq_run
    exe = os.path.join(ROOT, "apps", "bin", "vo_complete")
    a, b = str(tmp_path / "a") + "/", str(tmp_path / "b") + "/"
    os.makedirs(a); os.makedirs(b)
    subprocess.run([exe, d, a, str(ROUNDS)], check=True, stdout=subprocess.DEVNULL, timeout=600)
    subprocess.run([exe, d, b, str(ROUNDS), "--resident"], check=True, stdout=subprocess.DEVNULL, timeout=600)
    # ... and with every consecutive pair matched by one batched call before the chain (DeviceSequence::setMatchUpFront)
    c = str(tmp_path / "c") + "/"
    os.makedirs(c)
    out_b = subprocess.run([exe, d, c, str(ROUNDS), "--resident", "--match-up-front"], check=True, capture_output=True, text=True,
                           timeout=600).stdout
    assert " matches, " in out_b
    for other in (b, c):
        for name in ("trajectory_est_complete.txt", "trajectory_est_data.txt", "map.txt", "map_appearances.txt"):
            x, y = np.loadtxt(a + name), np.loadtxt(other + name)
            assert x.shape == y.shape, name
            assert np.array_equal(x, y), (name, float(np.abs(x - y).max()))


@pytest.mark.parametrize("case", ["normal", "empty5", "foreign5", "tiny5", "empty_last"])
def test_degenerate_sequences_follow_the_reference_loop(vo, ctx, o32, case):
    """Tracking lost in the middle of a sequence: a frame without measurements, a frame whose appearances match nothing, a
    frame of three points, an empty last frame.  The reference's loop has no special case for any of them (zero
    correspondences leave the pose at the identity, a zero baseline makes the next triangulation degenerate, NaN points
    poison the poses after that); the device-resident chain, with the solver in reference-order arithmetic, must do
    exactly the same: counts equal, finite poses bit-identical, NaN where the oracle's loop has NaN."""
    from oracle import vo_pipeline as vp
    seq = vo.synth.sequence(seed=3100, n_frames=12, n_visible=600)
    fr = seq["frames"]
    empty = dict(ids=np.zeros(0, np.int64), pts=np.zeros((0, 2), np.float32), app=np.zeros((0, 10), np.float32))
    if case == "empty5":
        fr[5] = empty
    if case == "empty_last":
        fr[11] = empty
    if case == "foreign5":
        fr[5] = dict(ids=fr[5]["ids"], pts=fr[5]["pts"], app=np.random.default_rng(1).uniform(5, 6, fr[5]["app"].shape).astype(np.float32))
    if case == "tiny5":
        fr[5] = dict(ids=fr[5]["ids"][:3], pts=fr[5]["pts"][:3].copy(), app=fr[5]["app"][:3].copy())
    sp = vo.SequencePipeline(ctx, seq, n_iters=30, exact=True)
    sp.run()
    traj, counts = sp.trajectory(), sp.counts()
    sp.close()
    res = vp.run_sequence([(f["pts"], f["app"]) for f in fr], seq["K"], seq["rows"], seq["cols"], seq["z_near"], seq["z_far"], 30, o32,
                          X0=traj[1], keep_map=False)
    ref = np.array(res["trajectory"], np.float32)
    assert np.array_equal(counts[2:, :2], np.array([s[:2] for s in res["stats"]]))         # matches, joined pairs
    assert np.array_equal(counts[1:, 2], np.array(res["tri_counts"]))                       # triangulated points
    assert np.array_equal(np.isnan(traj), np.isnan(ref)) and np.array_equal(traj, ref, equal_nan=True)
    assert np.isfinite(ref).all() == (case in ("normal", "tiny5", "empty_last"))            # the lost frames do poison the rest
    # the same sequence with every pair matched up front by one ragged batched call: empty and three-point frames included
    sp = vo.SequencePipeline(ctx, seq, n_iters=30, exact=True, prematch=True)
    sp.run()
    assert np.array_equal(sp.counts(), counts) and np.array_equal(sp.trajectory(), traj, equal_nan=True)
    sp.close()


@pytest.mark.parametrize("case", ["empty5", "foreign5", "tiny5"])
def test_degenerate_sequences_in_the_default_arithmetic(vo, ctx, o32, case):
    """The same lost-track sequences through the DEFAULT solver mode, whose 6x6 solve does not pivot: with no usable
    correspondence H is the damping alone (the identity: damping is fixed at 1 like the reference's, picp_solver.cpp:10 --
    there is no setter on either side), with three it is rank-deficient plus the identity -- positive definite either way,
    which is all the unpivoted factorisation needs.  Up to the first frame that loses track the chain must follow the
    reference-order one (counts equal, poses within 1e-3); from there on NaN must appear where it appears there."""
    seq = vo.synth.sequence(seed=3100, n_frames=12, n_visible=600)
    fr = seq["frames"]
    if case == "empty5":
        fr[5] = dict(ids=np.zeros(0, np.int64), pts=np.zeros((0, 2), np.float32), app=np.zeros((0, 10), np.float32))
    if case == "foreign5":
        fr[5] = dict(ids=fr[5]["ids"], pts=fr[5]["pts"], app=np.random.default_rng(1).uniform(5, 6, fr[5]["app"].shape).astype(np.float32))
    if case == "tiny5":
        fr[5] = dict(ids=fr[5]["ids"][:3], pts=fr[5]["pts"][:3].copy(), app=fr[5]["app"][:3].copy())
    out = {}
    for exact in (True, False):
        sp = vo.SequencePipeline(ctx, seq, n_iters=30, exact=exact)
        sp.run()
        out[exact] = (sp.trajectory(), sp.counts())
        sp.close()
    (te, ce), (tf, cf) = out[True], out[False]
    assert np.array_equal(ce[:6], cf[:6])                                   # up to and including the degenerate frame: same counts
    assert np.abs(te[:5] - tf[:5]).max() < 1e-3                              # the healthy part of the chain
    assert np.isfinite(tf[5]).all() and np.isfinite(te[5]).all()             # the degenerate frame itself: a finite pose in both
    if case in ("empty5", "foreign5"):
        assert np.array_equal(tf[5], np.eye(4, dtype=np.float32))            # no correspondence: dx = 0, the pose stays the identity
    assert np.array_equal(np.isnan(tf).any(axis=(1, 2)), np.isnan(te).any(axis=(1, 2)))     # poisoned frames: the same ones


def test_device_map_inside_the_chain_equals_the_reference_loop(vo, ctx, o32, seq_run):
    """keep_map: map.update(history * triangulated_pc) and history = history * pose^-1 (vo_complete.cpp:145-147,175-176)
    on the device, inside the chain, solver in reference-order arithmetic: the chain is then the oracle's chain bit for bit
    (tests above), so the map must be the oracle's map -- every entry, in order, point and appearance bits -- and the history
    isometry the oracle's."""
    seq, d, res0, P = seq_run
    sp = vo.SequencePipeline(ctx, seq, n_iters=ROUNDS, exact=True, keep_map=True, map_capacity=64)     # (it has to grow)
    sp.run()
    traj = sp.trajectory()
    pts, app = sp.map.read()
    hist = sp.map.history()
    frames = [(f["pts"], f["app"]) for f in seq["frames"]]
    res = P.run_sequence(frames, seq["K"], seq["rows"], seq["cols"], seq["z_near"], seq["z_far"], rounds=ROUNDS, o=o32, X0=traj[1])
    assert np.array_equal(np.array(traj), np.array(res["trajectory"], dtype=np.float32))
    m = res["map"]
    assert len(pts) == len(m.pts) > 2 * N_VISIBLE
    assert pts.tobytes() == np.array(m.pts, np.float32).tobytes()
    assert app.tobytes() == np.array(m.app, np.float32).tobytes()
    h = P.iso_inv32(traj[1])
    for X in traj[2:]:
        h = P.iso_mul32(h, P.iso_inv32(X))
    assert np.array_equal(hist, h)
    # running it again on the same object starts from an empty map
    sp.run()
    p2, a2 = sp.map.read()
    assert p2.tobytes() == pts.tobytes() and a2.tobytes() == app.tobytes()
    sp.close()
    # the default arithmetic keeps the same entries (the appearances do not depend on the pose)
    sp = vo.SequencePipeline(ctx, seq, n_iters=ROUNDS, keep_map=True)
    sp.run()
    p3, a3 = sp.map.read()
    sp.close()
    assert a3.tobytes() == app.tobytes() and np.abs(p3 - pts).max() < 5e-2


def test_device_map_with_the_matcher_up_front_or_on_a_second_stream(vo, ctx, seq_run):
    """keep_map together with the other ways the chain can get its matches (all pairs by one batched call first; the
    matcher one frame ahead on a second stream): the same chain, hence the same map, entry for entry."""
    seq, d, res0, P = seq_run
    maps = []
    for kw in ({}, {"prematch": True}, {"overlap_match": True}):
        sp = vo.SequencePipeline(ctx, seq, n_iters=ROUNDS, keep_map=True, **kw)
        sp.run()
        p, a = sp.map.read()
        maps.append((sp.trajectory().tobytes(), p.tobytes(), a.tobytes(), sp.map.history().tobytes()))
        sp.close()
    assert maps[0] == maps[1] == maps[2] and len(maps[0][1]) > 12 * 2 * N_VISIBLE
