// vo_da_known -- counterpart of the reference's src/tests/vo_daKnown.cpp:54-168 on the GPU path: monocular VO on a data
// directory with the data association GIVEN (landmark ids): relative pose of the first two frames by epipolar geometry,
// triangulation, then projective ICP between subsequent frames on the triangulated points.
//   usage: vo_da_known <data dir> [output dir] [rounds=1000] [--exact]
// Writes trajectory_gt.txt, trajectory_est_noWorld.txt and time_known.txt (ms spent in the id-based association per
// frame) like the reference, plus poses_raw.txt.  Prints per frame the counts and, at the end, the median translation
// ratio against trajectory.dat (the estimate is defined up to scale).
#include <chrono>
#include <fstream>

#include "known_common.hpp"

using namespace vo;
using namespace known;

static double getTime() {   // utils.cpp:2-6, milliseconds
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char* argv[]) {
  const Args a = parse(argc, argv, 1000);
  if (!a.ok) return -1;
  try {
    save_gt_trajectory(a.path + "trajectory.dat", a.out + "trajectory_gt.txt");
    const std::regex pattern("^meas-\\d.*\\.dat$");
    std::set<std::string> files;
    if (!get_file_names(a.path, files, pattern)) { std::cout << "unable to open directory\n"; return -1; }
    if (files.size() < 2) { std::cout << "need at least two measurement files\n"; return -1; }
    const auto first_file = *(files.begin());
    const auto second_file = *(files.erase(files.begin()));
    files.erase(files.begin());

    Vector3fVector reference_image_points_withid, current_image_points_withid;   // (landmark id, col, row)
    Vector10fVector reference_appearances, current_appearances;
    if (!get_meas_content(a.path + first_file, reference_appearances, reference_image_points_withid)) { std::cout << "Unable to open file 1\n"; return -1; }
    if (!get_meas_content(a.path + second_file, current_appearances, current_image_points_withid)) { std::cout << "Unable to open file 2\n"; return -1; }
    // the pairs are (ref_idx,curr_idx)
    IntPairVector correspondences_imgs = extract_correspondences_images(reference_image_points_withid, current_image_points_withid);
    Vector2fVector reference_image_points = strip_id(reference_image_points_withid);
    Vector2fVector current_image_points = strip_id(current_image_points_withid);
    std::vector<int> int_params;   // z_near,z_far,cols,rows
    Matrix3f k;
    Isometry3f H;
    if (!get_camera_params(a.path + "camera.dat", int_params, k, H)) { std::cout << "Unable to get camera parameters\n"; return -1; }
    Camera cam(int_params[3], int_params[2], int_params[0], int_params[1], k);

    const Isometry3f X = estimate_transform(cam.cameraMatrix(), correspondences_imgs, reference_image_points, current_image_points);
    Vector3fVector triangulated;
    IntPairVector correspondences_world;
    triangulate_points(k, X, correspondences_imgs, reference_image_points, current_image_points, triangulated, correspondences_world);   // (curr_idx,world_idx)
    // X is the pose of frame 0 in frame 1; "triangulated" are points expressed in frame 0

    IsometryVector trajectory;
    trajectory.reserve(files.size() + 2);
    trajectory.push_back(Isometry3f::Identity());
    trajectory.push_back(X);
    PICPSolver solver;
    solver.setKernelThreshold(10000);
    solver.setExact(a.exact);
    Isometry3f X_curr = X;
    reference_image_points = current_image_points;
    reference_image_points_withid = current_image_points_withid;   // correspondences_world now reads (ref_idx,world_idx)
    std::ofstream time_file(a.out + "time_known.txt");
    for (const auto& file : files) {
      if (!get_meas_content(a.path + file, current_appearances, current_image_points_withid)) { std::cout << "Unable to open file " << a.path + file << std::endl; return -1; }
      const double t_start = getTime();
      correspondences_imgs = extract_correspondences_images(reference_image_points_withid, current_image_points_withid);
      const double t_end = getTime();
      current_image_points = strip_id(current_image_points_withid);
      reference_image_points = strip_id(reference_image_points_withid);
      correspondences_world = extract_correspondences_world(correspondences_imgs, correspondences_world);
      triangulated = transform_points(X_curr, triangulated);         // for (auto& p : triangulated) p = X_curr * p;
      cam.setWorldInCameraPose(Isometry3f::Identity());
      solver.init(cam, triangulated, current_image_points);           // finds the current pose in the frame of the previous
      for (int i = 0; i < a.rounds; i++) solver.oneRound(correspondences_world, false);
      cam = solver.camera();
      trajectory.push_back(cam.worldInCameraPose());
      X_curr = cam.worldInCameraPose();
      std::printf("%s: %zu associated, %zu model correspondences, %d inliers, t = % .5f % .5f % .5f\n", file.c_str(), correspondences_imgs.size(),
                  correspondences_world.size(), solver.numInliers(), X_curr(0, 3), X_curr(1, 3), X_curr(2, 3));
      triangulate_points(k, cam.worldInCameraPose(), correspondences_imgs, reference_image_points, current_image_points, triangulated, correspondences_world);
      reference_image_points = current_image_points;
      reference_image_points_withid = current_image_points_withid;
      time_file << t_end - t_start << std::endl;
    }
    time_file.close();
    save_trajectory(a.out + "trajectory_est_noWorld.txt", trajectory, H);
    write_poses_raw(a.out + "poses_raw.txt", trajectory);

    // up to scale: the evaluation's median translation ratio (evaluate.cpp:40-60) against the ground truth
    const IsometryVector gt = get_gt_data(a.path + "trajectory.dat");
    IsometryVector est;
    {
      Isometry3f W = Isometry3f::Identity();
      const Isometry3f Ci = H.inverse();
      for (const auto& T : trajectory) { W = W * H * T.inverse() * Ci; est.push_back(W); }
    }
    std::vector<float> ratio;
    for (size_t i = 1; i < est.size() && i < gt.size(); ++i) {
      const Isometry3f Xr = est[i - 1].inverse() * est[i], Xg = gt[i - 1].inverse() * gt[i];
      float nr = 0.f, ng = 0.f;
      for (int r = 0; r < 3; ++r) { nr += Xr(r, 3) * Xr(r, 3); ng += Xg(r, 3) * Xg(r, 3); }
      if (ng > 0.f) ratio.push_back(std::sqrt(nr) / std::sqrt(ng));
    }
    const float inv_ratio = ratio.empty() ? 0.f : 1.f / median(ratio);
    std::printf("median translation ratio, inverted: %.6f; max abs deviation from the ground truth after scaling: %.3g\n", inv_ratio,
                max_error_vs_gt(trajectory, H, gt, inv_ratio));
    return 0;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "vo_da_known: %s\n", e.what());
    return 2;
  }
}
