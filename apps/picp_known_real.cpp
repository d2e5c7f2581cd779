// picp_known_real -- counterpart of the reference's src/tests/picp_real_data_allKnown.cpp:26-91 on the GPU path:
// projective ICP on a data directory with the landmark positions (world.dat) AND the data association (the landmark
// ids of the measurement files) known.  Per frame the world points are moved by the last estimate, the solver starts
// from the identity and runs `rounds` rounds; the relative poses are composed into trajectory_est.txt.
//   usage: picp_known_real <data dir> [output dir] [rounds=1000] [--exact]
// With exact measurements every pose must come out at the ground truth of trajectory.dat: the program prints the
// largest deviation and exits 0 iff it is below 2e-4.  Also written: poses_raw.txt (as vo_complete).
#include "known_common.hpp"

using namespace vo;
using namespace known;

int main(int argc, char* argv[]) {
  const Args a = parse(argc, argv, 1000);
  if (!a.ok) return -1;
  try {
    save_gt_trajectory(a.path + "trajectory.dat", a.out + "trajectory_gt.txt");
    const std::regex pattern("^meas-\\d.*\\.dat$");
    std::set<std::string> files;
    if (!get_file_names(a.path, files, pattern)) { std::cout << "unable to open directory\n"; return -1; }
    Vector3fVector world_points;
    Vector10fVector world_points_appearances;
    if (!get_meas_content(a.path + "world.dat", world_points_appearances, world_points, true)) { std::cout << "Unable to open world file\n"; return -1; }
    std::vector<int> int_params;   // z_near,z_far,cols,rows
    Matrix3f k;
    Isometry3f H;
    if (!get_camera_params(a.path + "camera.dat", int_params, k, H)) { std::cout << "Unable to get camera parameters\n"; return -1; }

    Camera cam(int_params[3], int_params[2], int_params[0], int_params[1], k);
    IsometryVector trajectory;
    trajectory.reserve(files.size());
    PICPSolver solver;
    solver.setKernelThreshold(10000);
    solver.setExact(a.exact);
    Isometry3f X_curr = H.inverse();
    for (const auto& file : files) {
      Vector3fVector current_image_points_withid;
      Vector10fVector current_appearances;
      if (!get_meas_content(a.path + file, current_appearances, current_image_points_withid)) { std::cout << "Unable to open file " << a.path + file << std::endl; return -1; }
      world_points = transform_points(X_curr, world_points);          // for (auto& p : world_points) p = X_curr * p;
      const Vector2fVector current_image_points = strip_id(current_image_points_withid);
      const IntPairVector correspondences_world = computeFakeCorrespondencesWorld(current_image_points_withid);
      cam.setWorldInCameraPose(Isometry3f::Identity());
      solver.init(cam, world_points, current_image_points);           // finds the current pose in the frame of the previous
      for (int i = 0; i < a.rounds; i++) solver.oneRound(correspondences_world, false);
      cam = solver.camera();
      trajectory.push_back(cam.worldInCameraPose());
      X_curr = cam.worldInCameraPose();
      std::printf("%s: %zu correspondences, %d inliers, t = % .6f % .6f % .6f\n", file.c_str(), correspondences_world.size(),
                  solver.numInliers(), X_curr(0, 3), X_curr(1, 3), X_curr(2, 3));
    }
    save_trajectory(a.out + "trajectory_est.txt", trajectory, H);
    write_poses_raw(a.out + "poses_raw.txt", trajectory);
    const float err = max_error_vs_gt(trajectory, H, get_gt_data(a.path + "trajectory.dat"));
    std::printf("max abs deviation from the ground-truth trajectory: %.3g\n", err);
    return err < 2e-4f ? 0 : 1;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "picp_known_real: %s\n", e.what());
    return 2;
  }
}
