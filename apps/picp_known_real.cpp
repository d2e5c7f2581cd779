// picp_known_real -- the scenario of the reference's src/tests/picp_real_data_allKnown.cpp (:26-91) on the GPU path: projective
// ICP on a data directory whose landmark positions (world.dat) AND data association (the landmark id of every measurement)
// are known.  Frame by frame the landmark cloud is carried into the previous camera's frame, the solver starts from the
// identity and runs `rounds` rounds; the relative poses are composed into trajectory_est.txt.
//   usage: picp_known_real <data dir> [output dir] [rounds=1000] [--exact]
// With exact measurements every pose must come out at the ground truth of trajectory.dat: the program prints the largest
// deviation and exits 0 iff it is below 2e-4 -- this is what pins the path on the reference's own data.
// Also written: poses_raw.txt (as vo_complete).
#include "known_common.hpp"

namespace {
// one frame: (measurement index, landmark id) pairs straight from the ids, `rounds` Gauss-Newton rounds from the identity
vo::Isometry3f relative_pose(vo::PICPSolver& solver, const vo::Camera& cam0, const vo::Vector3fVector& cloud_in_prev,
                             const vo::Vector3fVector& meas_with_id, int rounds, size_t& n_assoc) {
  const vo::IntPairVector assoc = known::computeFakeCorrespondencesWorld(meas_with_id);
  n_assoc = assoc.size();
  vo::Camera cam = cam0;
  cam.setWorldInCameraPose(vo::Isometry3f::Identity());
  solver.init(cam, cloud_in_prev, known::strip_id(meas_with_id));
  for (int r = 0; r < rounds; ++r) solver.oneRound(assoc, false);
  return solver.camera().worldInCameraPose();
}
}  // namespace

int main(int argc, char* argv[]) {
  const known::Args args = known::parse(argc, argv, 1000);
  if (!args.ok) return -1;
  try {
    known::Dataset data;
    if (!known::load_dataset(args.path, data, true)) return -1;
    vo::save_gt_trajectory(args.path + "trajectory.dat", args.out + "trajectory_gt.txt");
    vo::PICPSolver solver;
    solver.setKernelThreshold(10000);
    solver.setExact(args.exact);
    const vo::Camera cam0 = data.camera();
    vo::IsometryVector relative;
    relative.reserve(data.meas_files.size());
    vo::Vector3fVector cloud = data.landmarks;                     // the map, expressed in the frame of the last camera
    vo::Isometry3f last = data.cameraInRobot.inverse();            // world -> first camera
    for (const std::string& name : data.meas_files) {
      vo::Vector3fVector meas_with_id;
      vo::Vector10fVector appearances;
      if (!vo::get_meas_content(args.path + name, appearances, meas_with_id)) { std::printf("unable to read %s\n", (args.path + name).c_str()); return -1; }
      cloud = vo::transform_points(last, cloud);
      size_t n_assoc = 0;
      last = relative_pose(solver, cam0, cloud, meas_with_id, args.rounds, n_assoc);
      relative.push_back(last);
      std::printf("%s: %zu correspondences, %d inliers, t = % .6f % .6f % .6f\n", name.c_str(), n_assoc, solver.numInliers(), last(0, 3),
                  last(1, 3), last(2, 3));
    }
    vo::save_trajectory(args.out + "trajectory_est.txt", relative, data.cameraInRobot);
    known::write_poses_raw(args.out + "poses_raw.txt", relative);
    const float err = known::max_error_vs_gt(relative, data.cameraInRobot, vo::get_gt_data(args.path + "trajectory.dat"));
    std::printf("max abs deviation from the ground-truth trajectory: %.3g\n", err);
    return err < 2e-4f ? 0 : 1;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "picp_known_real: %s\n", e.what());
    return 2;
  }
}
