// real_init -- counterpart of the reference's src/tests/initialization_real_data.cpp:35-102 on the GPU path: relative
// pose of the first two frames of a data directory from the KNOWN association (landmark ids) by the eight-point
// algorithm, then the triangulated points mapped by the camera-in-robot transform into the world frame.
//   usage: real_init <data dir> [output dir]
// Writes world.txt and triangulated.txt like the reference.  The true answer is in the data: the rotation and the
// direction of the translation of trajectory.dat's first step, and -- after scaling by |t_gt| / |t_est| -- the landmarks
// of world.dat.  The program prints the three deviations and exits 0 iff rotation < 1e-4, direction < 1e-3 and the
// median landmark error < 5e-3.
#include <algorithm>

#include "known_common.hpp"

using namespace vo;
using namespace known;

int main(int argc, char* argv[]) {
  const Args a = parse(argc, argv, 0);
  if (!a.ok) return -1;
  try {
    const std::regex pattern("^meas-\\d.*\\.dat$");
    std::set<std::string> files;
    if (!get_file_names(a.path, files, pattern)) { std::cout << "unable to open directory\n"; return -1; }
    if (files.size() < 2) { std::cout << "need at least two measurement files\n"; return -1; }
    const auto first_file = *(files.begin());
    const auto second_file = *(files.erase(files.begin()));
    Vector3fVector reference_image_points_withid, current_image_points_withid, world_points;
    Vector10fVector reference_appearances, current_appearances, world_points_appearances;
    if (!get_meas_content(a.path + first_file, reference_appearances, reference_image_points_withid)) { std::cout << "Unable to open file 1\n"; return -1; }
    if (!get_meas_content(a.path + second_file, current_appearances, current_image_points_withid)) { std::cout << "Unable to open file 2\n"; return -1; }
    if (!get_meas_content(a.path + "world.dat", world_points_appearances, world_points, true)) { std::cout << "Unable to open world file\n"; return -1; }
    write_eigen_vectors_to_file(a.out + "world.txt", world_points);
    // the pair is (ref_idx,curr_idx)
    const IntPairVector correspondences_imgs = extract_correspondences_images(reference_image_points_withid, current_image_points_withid);
    const Vector2fVector reference_image_points = strip_id(reference_image_points_withid);
    const Vector2fVector current_image_points = strip_id(current_image_points_withid);
    std::vector<int> int_params;   // z_near,z_far,cols,rows
    Matrix3f k;
    Isometry3f H;
    if (!get_camera_params(a.path + "camera.dat", int_params, k, H)) { std::cout << "Unable to get camera parameters\n"; return -1; }
    Camera cam(int_params[3], int_params[2], int_params[0], int_params[1], k);

    const Isometry3f X = estimate_transform(cam.cameraMatrix(), correspondences_imgs, reference_image_points, current_image_points);
    std::printf("R estimated:\n");
    for (int r = 0; r < 3; ++r) std::printf("% .7f % .7f % .7f\n", X(r, 0), X(r, 1), X(r, 2));
    std::printf("t estimated: % .7f % .7f % .7f\n", X(0, 3), X(1, 3), X(2, 3));

    // triangulate the points to compare them with the true ones
    Vector3fVector triangulated;
    IntPairVector correspondences_world;   // (curr_idx, index of the triangulated point)
    triangulate_points(k, X, correspondences_imgs, reference_image_points, current_image_points, triangulated, correspondences_world);
    const Vector3fVector in_world = transform_points(H, triangulated);      // for (auto& p : triangulated) p = H * p;
    write_eigen_vectors_to_file(a.out + "triangulated.txt", in_world);

    // the known answer: frame 0 seen from frame 1, in camera coordinates
    const IsometryVector gt = get_gt_data(a.path + "trajectory.dat");
    if (gt.size() < 2) { std::cout << "no ground truth\n"; return -1; }
    const Isometry3f X_gt = H.inverse() * gt[1].inverse() * gt[0] * H;
    float n_est = 0.f, n_gt = 0.f, e_rot = 0.f, e_dir = 0.f;
    for (int r = 0; r < 3; ++r) { n_est += X(r, 3) * X(r, 3); n_gt += X_gt(r, 3) * X_gt(r, 3); }
    const float scale = std::sqrt(n_gt) / std::sqrt(n_est);
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) e_rot = std::fmax(e_rot, std::fabs(X(r, c) - X_gt(r, c)));
      e_dir = std::fmax(e_dir, std::fabs(X(r, 3) * scale - X_gt(r, 3)));
    }
    std::vector<float> e_pts;
    for (const auto& cw : correspondences_world) {
      const int id = (int)current_image_points_withid[(size_t)cw.first].x();
      if (id < 0 || id >= (int)world_points.size()) continue;
      // only the camera frame is scaled: H * (s * p_cam) = s * R p_cam + t
      const Vector3f& pc = triangulated[(size_t)cw.second];
      float d2 = 0.f;
      for (int r = 0; r < 3; ++r) {
        const float w = (H(r, 0) * pc[0] + H(r, 1) * pc[1] + H(r, 2) * pc[2]) * scale + H(r, 3);
        d2 += (w - world_points[(size_t)id][r]) * (w - world_points[(size_t)id][r]);
      }
      e_pts.push_back(std::sqrt(d2));
    }
    const float e_med = e_pts.empty() ? 1e9f : median(e_pts);
    std::printf("%zu correspondences, %zu triangulated; scale |t_gt|/|t_est| = %.6f\n", correspondences_imgs.size(), triangulated.size(), scale);
    std::printf("rotation error %.3g, translation direction error %.3g, median landmark error %.3g (max %.3g)\n", e_rot, e_dir, e_med,
                e_pts.empty() ? 0.f : *std::max_element(e_pts.begin(), e_pts.end()));
    return (e_rot < 1e-4f && e_dir < 1e-3f && e_med < 5e-3f) ? 0 : 1;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "real_init: %s\n", e.what());
    return 2;
  }
}
