// init_test -- counterpart of the reference's src/tests/initialization_test.cpp:43-89 on the GPU path: synthetic world seen
// from two poses, the relative pose estimated by epipolar geometry (estimate_transform: eight-point fundamental, essential
// matrix, cheirality vote on the GPU triangulation kernel) and printed next to the ground truth.
//   usage: init_test [seed=3] [n_points=1000]
// The translation is defined up to scale: the three ratios t_est/t_gt must agree.  exit code 0 iff the rotation is within 1e-4
// of the ground truth and the ratios agree to 1e-3.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "synth.hpp"
#include "vo/vo.hpp"

using namespace vo;

static void computeFakeCorrespondences(IntPairVector& correspondences, const Vector2fVector& reference_image_points,
                                       const Vector2fVector& current_measurements) {
  correspondences.clear();
  for (size_t i = 0; i < reference_image_points.size(); i++) {
    if (reference_image_points[i].x() < 0 || current_measurements[i].x() < 0) continue;   // the invalid point
    correspondences.push_back(IntPair((int)i, (int)i));
  }
}

int main(int argc, char** argv) {
  const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 3;
  const int n_points = argc > 2 ? atoi(argv[2]) : 1000;
  try {
    synth::Rng rng(seed);
    const Isometry3f X_gt = synth::generate_isometry3f(rng, 0.2f, 0.5f);
    const Vector3fVector world_points = synth::generate_points3d(rng, n_points);
    const Matrix3f k = Matrix3f::FromRows(150.f, 0.f, 320.f, 0.f, 150.f, 240.f, 0.f, 0.f, 1.f);
    Camera cam(480, 640, 0, 10, k);
    Vector2fVector reference_image_points, current_measurements;
    // since we keep indices, the i-th projection is the i-th world point
    cam.projectPoints(reference_image_points, world_points, true);
    cam.setWorldInCameraPose(X_gt);
    cam.projectPoints(current_measurements, world_points, true);
    IntPairVector correspondences;
    computeFakeCorrespondences(correspondences, reference_image_points, current_measurements);
    if (correspondences.size() < 8) { std::printf("only %zu points in both views: another seed\n", correspondences.size()); return 3; }

    const Isometry3f X_est = estimate_transform(cam.cameraMatrix(), correspondences, reference_image_points, current_measurements);
    std::printf("%zu correspondences\nR estimated | R gt\n", correspondences.size());
    float err_R = 0.f;
    for (int r = 0; r < 3; ++r) {
      std::printf("% .6f % .6f % .6f | % .6f % .6f % .6f\n", X_est(r, 0), X_est(r, 1), X_est(r, 2), X_gt(r, 0), X_gt(r, 1), X_gt(r, 2));
      for (int c = 0; c < 3; ++c) err_R = std::fmax(err_R, std::fabs(X_est(r, c) - X_gt(r, c)));
    }
    float ratio[3], lo = 1e30f, hi = -1e30f;
    std::printf("t ratio: ");
    for (int i = 0; i < 3; i++) {
      ratio[i] = X_est(i, 3) / X_gt(i, 3);
      std::printf("%.6f, ", ratio[i]);
      if (std::fabs(X_gt(i, 3)) > 0.02f) { lo = std::fmin(lo, ratio[i]); hi = std::fmax(hi, ratio[i]); }     // a near-zero component has no ratio
    }
    std::printf("\nmax |R - R_gt| %.3g, spread of the ratios %.3g\n", err_R, hi - lo);
    return (err_R < 1e-4f && lo > 0.f && hi - lo < 1e-3f * hi) ? 0 : 1;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "init_test: %s\n", e.what());
    return 2;
  }
}
