// picp_test -- counterpart of the reference's picp_solver_test.cpp:42-79 on the
// GPU solver: synthetic world, two poses, measurements by projectPoints with
// keep_indices, correspondences (i,i) where both views are valid, 1000 rounds
// from the identity, estimate printed next to the ground truth.
//   usage: picp_test [seed=1009] [n_points=1000] [rounds=1000]
// exit code 0 iff the estimate is within 1e-3 of the ground truth.
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
  const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 7;
  const int n_points = argc > 2 ? atoi(argv[2]) : 1000;
  const int rounds = argc > 3 ? atoi(argv[3]) : 1000;
  try {
    synth::Rng rng(seed);
    const Isometry3f X_gt = synth::generate_isometry3f(rng, 0.3f, 0.5f);
    const Vector3fVector world_points = synth::generate_points3d(rng, n_points);
    const Matrix3f k = Matrix3f::FromRows(180.f, 0.f, 320.f, 0.f, 180.f, 240.f, 0.f, 0.f, 1.f);
    Camera cam(480, 640, 0, 10, k);
    Vector2fVector reference_image_points, current_measurements;
    cam.projectPoints(reference_image_points, world_points, true);
    cam.setWorldInCameraPose(X_gt);
    cam.projectPoints(current_measurements, world_points, true);
    IntPairVector correspondences;
    computeFakeCorrespondences(correspondences, reference_image_points, current_measurements);
    cam.setWorldInCameraPose(Isometry3f::Identity());

    PICPSolver solver;
    solver.setKernelThreshold(10000);
    solver.init(cam, world_points, current_measurements);
    for (int i = 0; i < rounds; i++) solver.oneRound(correspondences, false);
    cam = solver.camera();

    const Isometry3f& X = cam.worldInCameraPose();
    float err = 0.f;
    std::printf("PICP solver: %zu correspondences, %d inliers, chi %.4g\nR estimated | R gt\n", correspondences.size(),
                solver.numInliers(), solver.chiInliers());
    for (int r = 0; r < 3; ++r) {
      std::printf("% .6f % .6f % .6f | % .6f % .6f % .6f\n", X(r, 0), X(r, 1), X(r, 2), X_gt(r, 0), X_gt(r, 1), X_gt(r, 2));
    }
    std::printf("t_est: % .6f % .6f % .6f\nt_gt : % .6f % .6f % .6f\n", X(0, 3), X(1, 3), X(2, 3), X_gt(0, 3), X_gt(1, 3), X_gt(2, 3));
    for (int i = 0; i < 16; ++i) err = std::fmax(err, std::fabs(X.m[i] - X_gt.m[i]));
    std::printf("max abs error %.3g\n", err);
    return err < 1e-3f ? 0 : 1;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "picp_test: %s\n", e.what());
    return 2;
  }
}
