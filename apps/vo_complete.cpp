// vo_complete -- counterpart of the reference's src/apps/vo_complete.cpp:68-187
// on the GPU path: monocular VO on a data directory with unknown data
// association.  Same call sequence, same output files (written into the
// current directory or into argv[2]): world.txt, trajectory_gt.txt, map.txt,
// map_appearances.txt, trajectory_est_complete.txt, trajectory_est_data.txt.
//   usage: vo_complete <data dir> [output dir] [rounds=100] [--resident [--match-up-front]] [--exact]
// --resident: the same sequence through vo::DeviceSequence -- all measurement files are read and uploaded first, the
// whole frame chain runs on the GPU without a host round trip per frame, the map upkeep included (vo_map_*: the
// reference's upsert as a hash table of first occurrences in device memory).  Same outputs.
// --exact: the solver in reference-order arithmetic (PICPSolver::setExact): every pose of the chain is then
// bit-identical to the reference's float32 arithmetic given the same first relative pose.
// Also written: poses_raw.txt, one camera pose per line (row-major 4x4, %.9g = exact float32 round trip), and map_raw.txt,
// one map entry per line (x y z a0..a9, %.9g).
#include <cstdio>
#include <iostream>

#include "vo/vo.hpp"

using namespace vo;

static void write_poses_raw(const std::string& file, const IsometryVector& trajectory) {
  std::FILE* f = std::fopen(file.c_str(), "w");
  if (!f) return;
  for (const auto& X : trajectory) {
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) std::fprintf(f, "%.9g ", X(r, c));
    std::fprintf(f, "\n");
  }
  std::fclose(f);
}

static void write_map_raw(const std::string& file, const PointCloudVector<3>& map) {
  std::FILE* f = std::fopen(file.c_str(), "w");
  if (!f) return;
  for (size_t i = 0; i < map.size(); ++i) {
    for (int k = 0; k < 3; ++k) std::fprintf(f, "%.9g ", map.points()[i][k]);
    for (int k = 0; k < 10; ++k) std::fprintf(f, "%.9g ", map.appearances()[i].v[k]);
    std::fprintf(f, "\n");
  }
  std::fclose(f);
}

// the device-resident form of the loop below: same call sequence, the frame chain inside vo::DeviceSequence
static int run_resident(const std::string& path, const std::string& out, int rounds, bool exact, bool up_front, const std::string& first_file,
                        const std::string& second_file, const std::set<std::string>& files) {
  std::vector<PointCloudVector<2>> frames;
  std::vector<std::string> names{first_file, second_file};
  names.insert(names.end(), files.begin(), files.end());
  for (const auto& f : names) {
    PointCloudVector<2> pc;
    if (!get_meas_content(path + f, pc)) { std::cout << "Unable to open file " << path + f << std::endl; return -1; }
    frames.push_back(std::move(pc));
  }
  Vector3fVector world_points;
  Vector10fVector world_points_appearances;
  if (!get_meas_content(path + "world.dat", world_points_appearances, world_points, true)) { std::cout << "Unable to open world file\n"; return -1; }
  write_eigen_vectors_to_file(out + "world.txt", world_points);
  std::vector<int> int_params;
  Matrix3f k;
  Isometry3f H;
  if (!get_camera_params(path + "camera.dat", int_params, k, H)) { std::cout << "Unable to get camera parameters\n"; return -1; }
  Camera cam(int_params[3], int_params[2], int_params[0], int_params[1], k);
  DeviceSequence seq(cam, frames, rounds);
  seq.setExact(exact);
  seq.setMatchUpFront(up_front);              // all consecutive pairs in one batched matcher call before the chain
  seq.setKeepMap(true);                       // map.update / history inside the chain, on the device (vo_complete.cpp:145-147,175-176)
  seq.run();
  const IsometryVector trajectory = seq.trajectory();          // waits for the chain
  for (int t = 2; t < seq.frames(); ++t) {
    int n_match, n_join, n_tri;
    seq.counts(t, n_match, n_join, n_tri);
    const Isometry3f& X = trajectory[(size_t)t];
    std::printf("%s: %d matches, %d model correspondences, t = % .5f % .5f % .5f\n", names[(size_t)t].c_str(), n_match, n_join,
                X(0, 3), X(1, 3), X(2, 3));
  }
  const PointCloudVector<3> map = seq.map(&H);                 // map = H * map (vo_complete.cpp:183)
  write_map_raw(out + "map_raw.txt", map);
  write_eigen_vectors_to_file(out + "map.txt", map.points());
  write_eigen_vectors_to_file(out + "map_appearances.txt", map.appearances());
  save_trajectory(out + "trajectory_est_complete.txt", trajectory, H);
  save_trajectory(out + "trajectory_est_data.txt", trajectory, H, true);
  write_poses_raw(out + "poses_raw.txt", trajectory);
  return 0;
}

int main(int argc, char* argv[]) {
  // flags first, wherever they stand; what is left are the positional arguments
  bool resident = false, exact = false, up_front = false;
  std::vector<std::string> pos;
  for (int i = 1; i < argc; ++i) {
    const std::string a(argv[i]);
    if (a == "--resident") resident = true;
    else if (a == "--exact") exact = true;
    else if (a == "--match-up-front") up_front = true;
    else if (a.rfind("--", 0) == 0) { std::cout << "unknown option " << a << std::endl; return -1; }
    else pos.push_back(a);
  }
  if (pos.empty()) { std::cout << "Error: need path parameter to read data" << std::endl; return -1; }
  std::string path(pos[0]);
  if (path.back() != '/') path.push_back('/');
  std::string out = pos.size() > 1 ? pos[1] : ".";
  if (out.back() != '/') out.push_back('/');
  const int rounds = pos.size() > 2 ? std::atoi(pos[2].c_str()) : 100;
  try {
    save_gt_trajectory(path + "trajectory.dat", out + "trajectory_gt.txt");
    const std::regex pattern("^meas-\\d.*\\.dat$");
    std::set<std::string> files;
    if (!get_file_names(path, files, pattern)) { std::cout << "unable to open directory\n"; return -1; }
    if (files.size() < 2) { std::cout << "need at least two measurement files\n"; return -1; }
    const auto first_file = *(files.begin());
    const auto second_file = *(files.erase(files.begin()));
    files.erase(files.begin());

    if (resident) return run_resident(path, out, rounds, exact, up_front, first_file, second_file, files);
    PointCloudVector<2> reference_pc, current_pc;
    if (!get_meas_content(path + first_file, reference_pc)) { std::cout << "Unable to open file measurement file 0\n"; return -1; }
    if (!get_meas_content(path + second_file, current_pc)) { std::cout << "Unable to open file measurement file 1\n"; return -1; }
    Vector3fVector world_points;
    Vector10fVector world_points_appearances;
    if (!get_meas_content(path + "world.dat", world_points_appearances, world_points, true)) { std::cout << "Unable to open world file\n"; return -1; }
    write_eigen_vectors_to_file(out + "world.txt", world_points);
    // the pairs are (ref_idx,curr_idx)
    IntPairVector correspondences_imgs = compute_correspondences_images(reference_pc.appearances(), current_pc.appearances());
    std::vector<int> int_params;   // z_near,z_far,cols,rows
    Matrix3f k;
    Isometry3f H;
    if (!get_camera_params(path + "camera.dat", int_params, k, H)) { std::cout << "Unable to get camera parameters\n"; return -1; }
    Camera cam(int_params[3], int_params[2], int_params[0], int_params[1], k);

    const Isometry3f X = estimate_transform(cam.cameraMatrix(), correspondences_imgs, reference_pc.points(), current_pc.points());

    PointCloudVector<3> triangulated_pc;
    IntPairVector correspondences_world;
    triangulate_points(k, X, correspondences_imgs, reference_pc, current_pc, triangulated_pc, correspondences_world);   // (curr_idx,world_idx)

    IsometryVector trajectory;
    trajectory.reserve(files.size() + 2);
    trajectory.push_back(Isometry3f::Identity());
    trajectory.push_back(X);
    PICPSolver solver;
    solver.setKernelThreshold(10000);
    solver.setExact(exact);

    reference_pc = current_pc;   // correspondences_world now reads (ref_idx,world_idx)
    PointCloudVector<3> triangulated_transformed, map;
    map.update(triangulated_pc);
    Isometry3f history = X.inverse();
    Isometry3f X_curr = X;

    for (const auto& file : files) {
      if (!get_meas_content(path + file, current_pc)) { std::cout << "Unable to open file " << path + file << std::endl; return -1; }
      correspondences_imgs = compute_correspondences_images(reference_pc.appearances(), current_pc.appearances());
      correspondences_world = extract_correspondences_world(correspondences_imgs, correspondences_world);
      triangulated_transformed = X_curr * triangulated_pc;
      cam.setWorldInCameraPose(Isometry3f::Identity());
      solver.init(cam, triangulated_transformed.points(), current_pc.points());
      for (int i = 0; i < rounds; i++) solver.oneRound(correspondences_world, false);
      cam = solver.camera();
      trajectory.push_back(cam.worldInCameraPose());
      X_curr = cam.worldInCameraPose();
      std::printf("%s: %zu matches, %zu model correspondences, %d inliers, t = % .5f % .5f % .5f\n", file.c_str(),
                  correspondences_imgs.size(), correspondences_world.size(), solver.numInliers(), X_curr(0, 3), X_curr(1, 3), X_curr(2, 3));
      triangulate_points(k, cam.worldInCameraPose(), correspondences_imgs, reference_pc, current_pc, triangulated_pc, correspondences_world);
      map.update(history * triangulated_pc);
      history = history * cam.worldInCameraPose().inverse();
      reference_pc = current_pc;
    }
    map = H * map;
    write_map_raw(out + "map_raw.txt", map);
    write_eigen_vectors_to_file(out + "map.txt", map.points());
    write_eigen_vectors_to_file(out + "map_appearances.txt", map.appearances());
    save_trajectory(out + "trajectory_est_complete.txt", trajectory, H);
    save_trajectory(out + "trajectory_est_data.txt", trajectory, H, true);
    write_poses_raw(out + "poses_raw.txt", trajectory);
    return 0;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "vo_complete: %s\n", e.what());
    return 2;
  }
}
