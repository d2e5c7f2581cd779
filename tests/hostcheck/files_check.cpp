// Host-only check of the facade's file readers / writers and of the evaluation (include/vo/files.hpp, vo/evaluation.hpp):
// reads a data directory, prints what it parsed as plain numbers, writes a trajectory with save_trajectory and evaluates
// files written by the test.  No GPU, no libvo_hip call.   usage: files_check <data dir> <work dir>
#include <cstdio>

#include "vo/evaluation.hpp"

using namespace vo;

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::string path(argv[1]), work(argv[2]);
  if (path.back() != '/') path.push_back('/');
  if (work.back() != '/') work.push_back('/');
  std::set<std::string> files;
  if (!get_file_names(path, files, std::regex("^meas-\\d.*\\.dat$"))) return 3;
  std::printf("files %zu first %s last %s\n", files.size(), files.begin()->c_str(), files.rbegin()->c_str());
  // both readers of a measurement file: (id, col, row) + appearance, and the point cloud form
  for (const auto& f : {*files.begin(), *files.rbegin()}) {
    Vector3fVector withid; Vector10fVector app; PointCloudVector<2> pc;
    if (!get_meas_content(path + f, app, withid) || !get_meas_content(path + f, pc)) return 4;
    double s_id = 0, s_uv = 0, s_app = 0, s_pc = 0;
    for (const auto& p : withid) { s_id += p[0]; s_uv += (double)p[1] + 2.0 * p[2]; }
    for (const auto& a : app) for (int k = 0; k < 10; ++k) s_app += (k + 1) * (double)a[k];
    for (size_t i = 0; i < pc.size(); ++i) { s_pc += (double)pc.points()[i][0] + 2.0 * pc.points()[i][1]; for (int k = 0; k < 10; ++k) s_pc += (k + 1) * (double)pc.appearances()[i][k]; }
    std::printf("meas %s n %zu ids %.17g uv %.17g app %.17g pc %zu %.17g\n", f.c_str(), withid.size(), s_id, s_uv, s_app, pc.size(), s_pc);
  }
  Vector3fVector world; Vector10fVector world_app;
  if (!get_meas_content(path + "world.dat", world_app, world, true)) return 5;
  double s_w = 0, s_wa = 0;
  for (const auto& p : world) s_w += (double)p[0] + 2.0 * p[1] + 3.0 * p[2];
  for (const auto& a : world_app) for (int k = 0; k < 10; ++k) s_wa += (k + 1) * (double)a[k];
  std::printf("world n %zu xyz %.17g app %.17g\n", world.size(), s_w, s_wa);
  std::vector<int> ip; Matrix3f k; Isometry3f H;
  if (!get_camera_params(path + "camera.dat", ip, k, H)) return 6;
  std::printf("camera ints %d %d %d %d K", ip[0], ip[1], ip[2], ip[3]);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) std::printf(" %.9g", k(r, c));
  std::printf(" H");
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) std::printf(" %.9g", H(r, c));
  std::printf("\n");
  const IsometryVector gt = get_gt_data(path + "trajectory.dat");
  double s_gt = 0;
  for (const auto& X : gt) for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) s_gt += (r * 4 + c + 1) * (double)X(r, c);
  std::printf("gt n %zu sum %.17g\n", gt.size(), s_gt);
  if (!save_gt_trajectory(path + "trajectory.dat", work + "trajectory_gt.txt")) return 7;
  // camera poses written by the test (one per line, row-major 4x4) -> save_trajectory in both forms
  IsometryVector traj;
  {
    std::ifstream in(work + "poses_in.txt");
    std::string line;
    while (std::getline(in, line)) {
      if (line.empty()) continue;
      std::stringstream ss(line);
      Isometry3f X = Isometry3f::Identity();
      for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ss >> X(r, c);
      traj.push_back(X);
    }
  }
  save_trajectory(work + "trajectory_est_complete.txt", traj, H);
  save_trajectory(work + "trajectory_est_data.txt", traj, H, true);
  // evaluation of those files + the map files written by the test
  const IsometryVector est = get_est_data(work + "trajectory_est_data.txt");
  const Vector3fVector map_est = read_eigen_vectors<3>(work + "map.txt");
  const Vector10fVector map_app = read_eigen_vectors<10>(work + "map_appearances.txt");
  const EvalResult r = evaluate(gt, est, map_est, map_app, world, world_app, work + "out_performance.txt");
  std::printf("eval n_est %zu e_theta %.9g inv_ratio %.9g rmse_pos %.9g rmse_map %.9g matched %d\n", est.size(), r.mean_orientation_error,
              r.median_ratio_inv, r.rmse_position, r.rmse_map, r.matched_map_points);
  return 0;
}
