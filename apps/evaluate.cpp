// evaluate -- counterpart of the reference's src/apps/evaluate.cpp: reads the
// files vo_complete wrote plus the ground truth and prints the README metrics.
//   usage: evaluate <data dir> [dir with vo_complete's outputs]
#include <cstdio>

#include "vo/evaluation.hpp"

using namespace vo;

int main(int argc, char* argv[]) {
  if (argc < 2) { std::cout << "Error: need path parameter to read data" << std::endl; return -1; }
  std::string path(argv[1]);
  if (path.back() != '/') path.push_back('/');
  std::string out = argc > 2 ? argv[2] : ".";
  if (out.back() != '/') out.push_back('/');
  const IsometryVector gt = get_gt_data(path + "trajectory.dat");
  const IsometryVector est = get_est_data(out + "trajectory_est_data.txt");
  const Vector3fVector map_est = read_eigen_vectors<3>(out + "map.txt");
  const Vector10fVector map_app = read_eigen_vectors<10>(out + "map_appearances.txt");
  Vector3fVector world;
  Vector10fVector world_app;
  if (!get_meas_content(path + "world.dat", world_app, world, true)) { std::cout << "Unable to open world file\n"; return -1; }
  if (gt.empty() || est.empty()) return -1;
  const EvalResult r = evaluate(gt, est, map_est, map_app, world, world_app, out + "out_performance.txt");
  std::printf("mean orientation error: %.6g\n", r.mean_orientation_error);
  std::printf("ratio used for map correction: %.6g\n", r.median_ratio_inv);
  std::printf("RMSE position: %.6g\n", r.rmse_position);
  std::printf("RMSE map: %.6g (%d points)\n", r.rmse_map, r.matched_map_points);
  return 0;
}
