// read_data_test -- counterpart of the reference's src/tests/read_data_test.cpp: the readers of the data directory, printed.
//   usage: read_data_test <data dir>
// Host code only (no GPU call).  exit code 0 iff every file parsed and the counts are consistent.
#include "vo/vo.hpp"

using namespace vo;

int main(int argc, char* argv[]) {
  if (argc < 2) { std::cout << "Error: need path parameter to read data" << std::endl; return -1; }
  std::string path(argv[1]);
  if (path.back() != '/') path.push_back('/');
  const std::regex pattern("^meas-\\d.*\\.dat$");
  std::set<std::string> files;
  if (!get_file_names(path, files, pattern)) { std::cout << "unable to open directory\n"; return -1; }
  std::cout << files.size() << " measurement files, first " << *files.begin() << ", last " << *files.rbegin() << std::endl;
  Vector3fVector features;           // (landmark id, col, row)
  Vector10fVector appearances;
  if (!get_meas_content(path + *files.begin(), appearances, features)) { std::cout << "Unable to open file\n"; return -1; }
  std::cout << *files.begin() << ": " << features.size() << " points\n";
  for (size_t i = 0; i < features.size() && i < 3; ++i) {
    std::cout << "  id " << features[i][0] << " col " << features[i][1] << " row " << features[i][2] << " appearance";
    for (int k = 0; k < 10; ++k) std::cout << " " << appearances[i][k];
    std::cout << "\n";
  }
  PointCloudVector<2> pc;
  if (!get_meas_content(path + *files.begin(), pc) || pc.size() != features.size()) { std::cout << "point-cloud reader disagrees\n"; return 1; }
  Vector3fVector world;
  Vector10fVector world_app;
  if (!get_meas_content(path + "world.dat", world_app, world, true)) { std::cout << "Unable to open world file\n"; return -1; }
  std::cout << "world.dat: " << world.size() << " landmarks, first " << world[0][0] << " " << world[0][1] << " " << world[0][2] << "\n";
  std::vector<int> int_params;   // z_near,z_far,cols,rows
  Matrix3f k;
  Isometry3f H;
  if (!get_camera_params(path + "camera.dat", int_params, k, H)) { std::cout << "Unable to get camera parameters\n"; return -1; }
  std::cout << "camera: z_near " << int_params[0] << " z_far " << int_params[1] << " cols " << int_params[2] << " rows " << int_params[3] << "\nK:\n";
  for (int r = 0; r < 3; ++r) std::cout << "  " << k(r, 0) << " " << k(r, 1) << " " << k(r, 2) << "\n";
  std::cout << "camera in robot:\n";
  for (int r = 0; r < 4; ++r) std::cout << "  " << H(r, 0) << " " << H(r, 1) << " " << H(r, 2) << " " << H(r, 3) << "\n";
  const IsometryVector gt = get_gt_data(path + "trajectory.dat");
  std::cout << "trajectory.dat: " << gt.size() << " poses, last at " << gt.back()(0, 3) << " " << gt.back()(1, 3) << "\n";
  const bool ok = gt.size() == files.size() && !world.empty() && int_params.size() == 4;
  return ok ? 0 : 1;
}
