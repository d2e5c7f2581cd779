"""Known answers HELD BY THE REFERENCE for the solver path: its data directory ships the ground truth (trajectory.dat,
world.dat, landmark ids in every measurement file), and its own test programs run the path on exactly these inputs:
  picp_known_real  (src/tests/picp_real_data_allKnown.cpp)  landmarks + association known -> every camera pose
  real_init        (src/tests/initialization_real_data.cpp) association known -> first relative pose, triangulated landmarks
  vo_daKnown       (src/tests/vo_daKnown.cpp)               association known -> whole trajectory up to scale
The measurements are exact projections, so the first two have a precise expected output -- the ground truth itself -- which
pins the ORACLE (file readers, rigid transform, projection, linearisation, 6x6 solve, v2tEuler, pose composition,
eight-point initialisation, triangulation) independently of any code of this repository.  tests/golden/example_data is the
reference's own data directory (data files, no code)."""
import os

import numpy as np
import pytest

from oracle import vo_pipeline as vp
from oracle.oracle import Oracle

DATA = os.path.join(os.path.dirname(__file__), "golden", "example_data", "data")


@pytest.mark.parametrize("bits", [32, 64])
def test_picp_with_known_landmarks_recovers_the_ground_truth_trajectory(bits):
    o = Oracle(bits)
    r = vp.run_picp_known_real(DATA, rounds=1000, o=o)
    assert len(r["trajectory"]) == 121
    err, _ = vp.gt_errors(DATA, r["trajectory"], r["H"])
    # 121 composed poses, 14..127 landmarks in view each; measured 4.7e-5 (float32 and float64 alike: the data files
    # carry six significant digits)
    assert err.max() < 1e-4, err.max()
    # exact measurements: every correspondence is an inlier, bar the one or two per frame that sit on the z_far / image gates
    assert all(n - 2 <= n_in <= n for n, n_in in r["stats"]) and sum(n == n_in for n, n_in in r["stats"]) >= 110
    # the reference runs 1000 rounds; the fixed point is reached long before
    r100 = vp.run_picp_known_real(DATA, rounds=100, o=o)
    assert np.abs(np.array(r100["trajectory"]) - np.array(r["trajectory"])).max() < 1e-5


def test_real_init_recovers_the_first_step_and_the_landmarks(o32):
    r = vp.run_real_init(DATA, o32)
    assert len(r["corr"]) == 115
    gt = vp.read_gt(os.path.join(DATA, "trajectory.dat"))
    H = r["H"].astype(np.float64)
    X_gt = np.linalg.inv(H) @ np.linalg.inv(gt[1]) @ gt[0] @ H          # frame 0 seen from frame 1, camera coordinates
    X = r["X"].astype(np.float64)
    scale = np.linalg.norm(X_gt[:3, 3]) / np.linalg.norm(X[:3, 3])
    assert np.abs(X[:3, :3] - X_gt[:3, :3]).max() < 2e-5                # measured 3.1e-6
    assert np.abs(X[:3, 3] * scale - X_gt[:3, 3]).max() < 5e-5         # measured 5.9e-6
    assert abs(scale - 0.47332) < 2e-4                                  # = the README's 1/r_t (0.47337) to 4 digits
    # triangulated landmarks, camera frame scaled, mapped by H: the landmarks of world.dat
    world, _ = vp.read_world(os.path.join(DATA, "world.dat"))
    cam_pts = (np.linalg.inv(H) @ np.c_[r["points"].astype(np.float64), np.ones(len(r["points"]))].T).T[:, :3]
    P = (H[:3, :3] @ (cam_pts * scale).T).T + H[:3, 3]
    e = np.linalg.norm(P - world[r["ids"]], axis=1)
    assert len(e) == 115 and np.median(e) < 2e-3 and e.max() < 0.1     # measured 4.8e-4 / 0.035 (depth 10)


def test_vo_with_known_association(o32):
    r = vp.run_vo_da_known(DATA, rounds=1000, o=o32)
    assert len(r["trajectory"]) == 121
    err, scale = vp.gt_errors(DATA, r["trajectory"], r["H"], up_to_scale=True)
    assert abs(scale - 0.47337) < 2e-4                                  # README 1/r_t; measured 0.473357
    assert err[:10].max() < 0.02 and err.max() < 0.5                    # monocular drift: measured 0.2 at the end
    # the id association and the appearance association are the same pairs on this data (appearances are unique)
    r_app = vp.run_vo_complete(DATA, rounds=1000, o=o32)
    assert [s[:2] for s in r["stats"]] == [s[:2] for s in r_app["stats"]]
    assert np.array_equal(np.array(r["trajectory"]), np.array(r_app["trajectory"]))


def test_id_association_is_the_ordered_merge():
    a = np.array([1, 4, 4, 7, 9]); b = np.array([0, 4, 7, 7, 10])
    assert vp.id_correspondences(a, b).tolist() == [[1, 1], [2, 1], [3, 2]]
    assert vp.id_correspondences(a, np.array([], int)).shape == (0, 2)
