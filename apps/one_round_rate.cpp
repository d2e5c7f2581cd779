// one_round_rate -- the reference's own call pattern, timed: vo_complete.cpp:160-168 is
//     solver.init(cam, points, measurements);
//     for (i < 100) solver.oneRound(correspondences_world, false);
//     cam = solver.camera();
// on vo::PICPSolver (include/vo/picp_solver.hpp -> vo_picp_one_round), at the size of BASELINE configs[1].
//   usage: one_round_rate [points=50000] [rounds=50] [steps=200] [warmup=20] [seed=2000]
// Prints one JSON object: iterations/s of the loop with the pose reset per step (`loop`) and with a full init() per
// step as the reference's frame loop does (`with_init`; `with_init_varying_sizes`: the vector's length changes from frame to
// frame), the host time per oneRound call (the calls alone, before anything waits), and what the closed entry point
// (solve: one call, all rounds) does on the same pair.
// exit code 0 iff every step's pose is the generator's motion (1e-3) with all correspondences inliers.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "synth.hpp"
#include "vo/vo.hpp"

using namespace vo;
using clk = std::chrono::steady_clock;

static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 50000;
  const int rounds = argc > 2 ? atoi(argv[2]) : 50;
  const int steps = argc > 3 ? atoi(argv[3]) : 200;
  const int warmup = argc > 4 ? atoi(argv[4]) : 20;
  const uint64_t seed = argc > 5 ? strtoull(argv[5], nullptr, 10) : 2000;
  try {
    synth::Rng g(seed);
    const Matrix3f k = Matrix3f::FromRows(180.f, 0.f, 320.f, 0.f, 180.f, 240.f, 0.f, 0.f, 1.f);
    const Isometry3f X_gt = synth::generate_isometry3f(g, 0.05f, 0.1f);
    // frustum-filling pair: n landmarks valid in both views, the current image in a random order
    Vector3fVector model((size_t)n);
    Vector2fVector meas((size_t)n);
    IntPairVector corr((size_t)n);
    std::vector<int> perm((size_t)n);
    std::iota(perm.begin(), perm.end(), 0);
    for (int i = n - 1; i > 0; --i) std::swap(perm[(size_t)i], perm[(size_t)(g.next() % (uint64_t)(i + 1))]);
    for (int i = 0; i < n;) {
      const float z = g.uniform(1.f, 9.f);
      const float x = g.uniform(-0.9f, 0.9f) * z * (319.5f / 180.f), y = g.uniform(-0.9f, 0.9f) * z * (239.5f / 180.f);
      const float xc = X_gt(0, 0) * x + X_gt(0, 1) * y + X_gt(0, 2) * z + X_gt(0, 3);
      const float yc = X_gt(1, 0) * x + X_gt(1, 1) * y + X_gt(1, 2) * z + X_gt(1, 3);
      const float zc = X_gt(2, 0) * x + X_gt(2, 1) * y + X_gt(2, 2) * z + X_gt(2, 3);
      const float u = 180.f * xc / zc + 320.f, v = 180.f * yc / zc + 240.f;
      if (zc < 0.5f || zc > 9.5f || u < 2 || u > 637 || v < 2 || v > 477) continue;
      model[(size_t)i][0] = x; model[(size_t)i][1] = y; model[(size_t)i][2] = z;
      meas[(size_t)perm[(size_t)i]][0] = u; meas[(size_t)perm[(size_t)i]][1] = v;
      corr[(size_t)i] = IntPair(perm[(size_t)i], i);         // (measurement, model): picp_solver.cpp:66-67
      ++i;
    }
    Camera cam(480, 640, 0, 10, k);
    PICPSolver solver;
    solver.setKernelThreshold(10000);
    solver.init(cam, model, meas);
    const Isometry3f I = Isometry3f::Identity();
    bool ok = true;
    auto verify = [&](const char* what) {
      const Isometry3f& X = solver.camera().worldInCameraPose();
      float err = 0.f;
      for (int i = 0; i < 16; ++i) err = std::fmax(err, std::fabs(X.m[i] - X_gt.m[i]));
      if (!(err < 1e-3f) || solver.numInliers() != n) { std::fprintf(stderr, "%s: pose error %g, %d inliers of %d\n", what, err, solver.numInliers(), n); ok = false; }
    };

    // (1) the loop: pose back to the identity, `rounds` oneRound calls, camera()
    double calls_us = 0;
    auto loop_step = [&](bool timed) {
      check(vo_picp_set_pose(solver.handle(), I.data()), "vo_picp_set_pose");
      const auto a = clk::now();
      for (int i = 0; i < rounds; ++i) solver.oneRound(corr, false);
      const auto b = clk::now();
      (void)solver.camera();
      if (timed) calls_us += us(a, b);
    };
    for (int s = 0; s < warmup; ++s) loop_step(false);
    auto t0 = clk::now();
    for (int s = 0; s < steps; ++s) loop_step(true);
    auto t1 = clk::now();
    const double loop_us = us(t0, t1);
    verify("loop");
    int open_rounds = 0;
    unsigned long long spec = 0, redone = 0;
    vo_picp_chain_info(solver.handle(), &open_rounds, &spec, &redone);

    // (2) the reference's frame: init() (camera + both point vectors to the GPU), the rounds, camera()
    auto init_step = [&] {
      solver.init(cam, model, meas);
      for (int i = 0; i < rounds; ++i) solver.oneRound(corr, false);
      (void)solver.camera();
    };
    for (int s = 0; s < warmup; ++s) init_step();
    t0 = clk::now();
    for (int s = 0; s < steps; ++s) init_step();
    t1 = clk::now();
    const double init_us = us(t0, t1);
    verify("with_init");

    // (3) the closed entry point on the same pair: one call, all rounds
    auto solve_step = [&] {
      check(vo_picp_set_pose(solver.handle(), I.data()), "vo_picp_set_pose");
      solver.solve(corr, false, rounds);
      (void)solver.camera();
    };
    for (int s = 0; s < warmup; ++s) solve_step();
    t0 = clk::now();
    for (int s = 0; s < steps; ++s) solve_step();
    t1 = clk::now();
    const double solve_us = us(t0, t1);
    verify("solve");

    // (4) the explicit hand-over: pairs given once, rounds with no comparison at all (what a call costs without it)
    check(vo_picp_set_correspondences(solver.handle(), pair_data(corr), n), "vo_picp_set_correspondences");
    double rcalls_us = 0;
    auto rounds_step = [&](bool timed) {
      check(vo_picp_set_pose(solver.handle(), I.data()), "vo_picp_set_pose");
      const auto a = clk::now();
      for (int i = 0; i < rounds; ++i) check(vo_picp_rounds(solver.handle(), 0, 1), "vo_picp_rounds");
      const auto b = clk::now();
      (void)solver.camera();
      if (timed) rcalls_us += us(a, b);
    };
    for (int s = 0; s < warmup; ++s) rounds_step(false);
    t0 = clk::now();
    for (int s = 0; s < steps; ++s) rounds_step(true);
    t1 = clk::now();
    const double rounds_us = us(t0, t1);
    verify("rounds_call");

    // (5) the reference's frame loop on frames of DIFFERENT sizes: init(), the rounds on a vector whose length changes from
    // frame to frame (n - 0..10 %), camera() -- what the launch bookkeeping costs when no two frames share a launch geometry
    std::vector<IntPairVector> sized;
    for (int k = 0; k < 37; ++k) sized.emplace_back(corr.begin(), corr.begin() + (n - (int)((long long)n * k / 370)));
    int fk = 0;
    auto varied_step = [&] {
      const IntPairVector& c = sized[(size_t)(fk++ % 37)];
      solver.init(cam, model, meas);
      for (int i = 0; i < rounds; ++i) solver.oneRound(c, false);
      (void)solver.camera();
    };
    for (int s = 0; s < warmup; ++s) varied_step();
    t0 = clk::now();
    for (int s = 0; s < steps; ++s) varied_step();
    t1 = clk::now();
    const double varied_us = us(t0, t1);
    {
      const Isometry3f& X = solver.camera().worldInCameraPose();
      float err = 0.f;
      for (int i = 0; i < 16; ++i) err = std::fmax(err, std::fabs(X.m[i] - X_gt.m[i]));
      if (!(err < 1e-3f)) { std::fprintf(stderr, "varied: pose error %g\n", err); ok = false; }
    }

    const double iters = (double)steps * rounds;
    std::printf("{\"points\": %d, \"rounds_per_step\": %d, \"steps\": %d, \"warmup\": %d, "
                "\"loop\": {\"iters_per_sec\": %.1f, \"us_per_round\": %.3f, \"host_us_per_call\": %.3f, \"ms_per_step\": %.4f}, "
                "\"with_init\": {\"iters_per_sec\": %.1f, \"us_per_round\": %.3f, \"ms_per_step\": %.4f}, "
                "\"solve_call\": {\"iters_per_sec\": %.1f, \"us_per_round\": %.3f, \"ms_per_step\": %.4f}, "
                "\"rounds_call\": {\"iters_per_sec\": %.1f, \"us_per_round\": %.3f, \"host_us_per_call\": %.3f}, "
                "\"with_init_varying_sizes\": {\"iters_per_sec\": %.1f, \"us_per_round\": %.3f, \"ms_per_step\": %.4f}, "
                "\"speculative_calls\": %llu, \"repeated_calls\": %llu, \"ok\": %s}\n",
                n, rounds, steps, warmup, iters / loop_us * 1e6, loop_us / iters, calls_us / iters, loop_us / steps * 1e-3,
                iters / init_us * 1e6, init_us / iters, init_us / steps * 1e-3, iters / solve_us * 1e6, solve_us / iters,
                solve_us / steps * 1e-3, iters / rounds_us * 1e6, rounds_us / iters, rcalls_us / iters,
                iters / varied_us * 1e6, varied_us / iters, varied_us / steps * 1e-3, spec, redone, ok ? "true" : "false");
    return ok ? 0 : 1;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "one_round_rate: %s\n", e.what());
    return 2;
  }
}
