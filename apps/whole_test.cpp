// whole_test -- counterpart of the reference's essential_picp_test.cpp:45-106:
// synthetic world seen from three poses; epipolar initialisation of cam0->cam1,
// triangulation, then PICP of a third view against the triangulated model.
// Monocular scale is free: the estimate is compared with X_gt2 * X_gt1^-1 up
// to the translation scale.   usage: whole_test [seed=3] [n_points=4000]
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "synth.hpp"
#include "vo/vo.hpp"

using namespace vo;

static void computeFakeCorrespondences(IntPairVector& c, const Vector2fVector& a, const Vector2fVector& b) {
  c.clear();
  for (size_t i = 0; i < a.size(); i++) if (!(a[i].x() < 0 || b[i].x() < 0)) c.push_back(IntPair((int)i, (int)i));
}

int main(int argc, char** argv) {
  const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 3;
  const int n_points = argc > 2 ? atoi(argv[2]) : 4000;
  try {
    synth::Rng rng(seed);
    const Isometry3f X_gt1 = synth::generate_isometry3f(rng, 0.2f, 0.5f);
    const Vector3fVector world_points_gt = synth::generate_points3d(rng, n_points);
    const Matrix3f k = Matrix3f::FromRows(150.f, 0.f, 320.f, 0.f, 150.f, 240.f, 0.f, 0.f, 1.f);
    Camera cam(480, 640, 0, 10, k);
    Vector2fVector reference_image_points, current_measurements;
    cam.projectPoints(reference_image_points, world_points_gt, true);
    cam.setWorldInCameraPose(X_gt1);
    cam.projectPoints(current_measurements, world_points_gt, true);
    IntPairVector correspondences;
    computeFakeCorrespondences(correspondences, reference_image_points, current_measurements);
    const Isometry3f X_est = estimate_transform(cam.cameraMatrix(), correspondences, reference_image_points, current_measurements);
    float scale = 0.f, err_R = 0.f;
    { float n1 = 0, n2 = 0; for (int i = 0; i < 3; ++i) { n1 += X_est(i, 3) * X_est(i, 3); n2 += X_gt1(i, 3) * X_gt1(i, 3); } scale = std::sqrt(n1 / n2); }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) err_R = std::fmax(err_R, std::fabs(X_est(r, c) - X_gt1(r, c)));
    float err_t = 0.f;
    for (int i = 0; i < 3; ++i) err_t = std::fmax(err_t, std::fabs(X_est(i, 3) / scale - X_gt1(i, 3)));
    std::printf("EPIPOLAR: %zu correspondences, |R - R_gt| %.3g, |t/s - t_gt| %.3g (scale %.4f)\n", correspondences.size(), err_R, err_t, scale);

    Vector3fVector world_points_est;
    IntPairVector correspondences_new;
    triangulate_points(k, X_est, correspondences, reference_image_points, current_measurements, world_points_est, correspondences_new);
    const Isometry3f X_gt2 = synth::generate_isometry3f(rng, 0.2f, 0.5f);
    cam.setWorldInCameraPose(X_gt2);
    cam.projectPoints(current_measurements, world_points_gt, true);
    // drop correspondences whose point is not visible from the third pose
    IntPairVector corr3;
    for (const IntPair& c : correspondences_new) if (!(current_measurements[(size_t)c.first].x() < 0)) corr3.push_back(c);
    PICPSolver solver;
    solver.setKernelThreshold(10000);
    const Vector3fVector points_in_cameraframe1 = transform_points(X_est, world_points_est);
    cam.setWorldInCameraPose(Isometry3f::Identity());
    solver.init(cam, points_in_cameraframe1, current_measurements);
    for (int i = 0; i < 100; i++) solver.oneRound(corr3, false);
    cam = solver.camera();
    const Isometry3f X_ref = X_gt2 * X_gt1.inverse();      // pose of camera 1 in camera 2
    const Isometry3f& X = cam.worldInCameraPose();
    float e_R = 0.f, e_t = 0.f;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) e_R = std::fmax(e_R, std::fabs(X(r, c) - X_ref(r, c)));
    for (int i = 0; i < 3; ++i) e_t = std::fmax(e_t, std::fabs(X(i, 3) / scale - X_ref(i, 3)));
    std::printf("PICP: %zu correspondences, %d inliers, |R - R_gt| %.3g, |t/s - t_gt| %.3g\n", corr3.size(), solver.numInliers(), e_R, e_t);
    // exact synthetic measurements: the eight-point initialisation must hit the generating motion (float32 pixel rounding only)
    return (err_R < 1e-4f && err_t < 1e-4f && e_R < 1e-2f && e_t < 5e-2f) ? 0 : 1;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "whole_test: %s\n", e.what());
    return 2;
  }
}
