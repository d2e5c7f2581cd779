// batch_frames -- BASELINE config 4 from C++: F independent synthetic frame pairs, generated here, resident in
// HBM, solved by ONE vo_frames_batch_dev call (match -> join -> transform -> n rounds -> triangulate for every
// frame, the frame as a grid dimension), poses checked against the generator's ground truth.
//   usage: batch_frames [frames=64] [points=20000] [rounds=50] [repeats=5]
// Plain C++ over the C ABI (include/vo_hip.h): no facade classes, no Python.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "synth.hpp"
#include "vo_hip.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    const int rc_ = (call);                                                              \
    if (rc_ != VO_OK) { std::fprintf(stderr, "%s: %s\n", #call, vo_last_error()); return 2; } \
  } while (0)

namespace {
struct Pair {
  std::vector<float> ref_app, cur_app, ref_pts, cur_pts, model;
  std::vector<int32_t> model_pairs;
  vo::Isometry3f X_gt;
};

// n landmarks visible in both views of a small motion X_gt (p_cur = X_gt p_ref), appearance copied exactly into
// both images, current image in a random order -- the frustum-filling pair of SURVEY 8(d) config 2
Pair make_pair(int n, uint64_t seed, const float K[9]) {
  synth::Rng g(seed);
  Pair p;
  p.X_gt = synth::generate_isometry3f(g, 0.05f, 0.1f);
  const float fx = K[0], fy = K[4], cx = K[6], cy = K[7];
  p.ref_app.resize(10 * (size_t)n); p.cur_app.resize(10 * (size_t)n);
  p.ref_pts.resize(2 * (size_t)n); p.cur_pts.resize(2 * (size_t)n);
  p.model.resize(3 * (size_t)n); p.model_pairs.resize(2 * (size_t)n);
  std::vector<int> perm((size_t)n);
  std::iota(perm.begin(), perm.end(), 0);
  for (int i = n - 1; i > 0; --i) std::swap(perm[(size_t)i], perm[(size_t)(g.next() % (uint64_t)(i + 1))]);
  int i = 0;
  while (i < n) {
    const float z = g.uniform(1.f, 9.f);
    const float x = g.uniform(-0.9f, 0.9f) * z * (319.5f / fx), y = g.uniform(-0.9f, 0.9f) * z * (239.5f / fy);
    // the point in the reference frame, and its image in the current camera
    const vo::Isometry3f& X = p.X_gt;
    const float xc = X(0, 0) * x + X(0, 1) * y + X(0, 2) * z + X(0, 3);
    const float yc = X(1, 0) * x + X(1, 1) * y + X(1, 2) * z + X(1, 3);
    const float zc = X(2, 0) * x + X(2, 1) * y + X(2, 2) * z + X(2, 3);
    const float u0 = fx * x / z + cx, v0 = fy * y / z + cy, u1 = fx * xc / zc + cx, v1 = fy * yc / zc + cy;
    if (zc < 0.5f || zc > 9.5f || u1 < 2 || u1 > 637 || v1 < 2 || v1 > 477 || u0 < 2 || u0 > 637 || v0 < 2 || v0 > 477) continue;
    const int j = perm[(size_t)i];
    p.model[3 * (size_t)i] = x; p.model[3 * (size_t)i + 1] = y; p.model[3 * (size_t)i + 2] = z;
    p.model_pairs[2 * (size_t)i] = i; p.model_pairs[2 * (size_t)i + 1] = i;
    p.ref_pts[2 * (size_t)i] = u0; p.ref_pts[2 * (size_t)i + 1] = v0;
    p.cur_pts[2 * (size_t)j] = u1; p.cur_pts[2 * (size_t)j + 1] = v1;
    for (int k = 0; k < 10; ++k) p.ref_app[10 * (size_t)i + k] = p.cur_app[10 * (size_t)j + k] = g.uniform(-1.f, 1.f);
    ++i;
  }
  return p;
}

template <class T>
T* upload(vo_ctx* ctx, const std::vector<T>& host) {
  void* d = nullptr;
  if (vo_dev_alloc(ctx, host.size() * sizeof(T), &d) != VO_OK) return nullptr;
  if (vo_memcpy_h2d(ctx, d, host.data(), host.size() * sizeof(T)) != VO_OK) return nullptr;
  return static_cast<T*>(d);
}
template <class T>
T* alloc(vo_ctx* ctx, size_t n) {
  void* d = nullptr;
  return vo_dev_alloc(ctx, n * sizeof(T), &d) == VO_OK ? static_cast<T*>(d) : nullptr;
}
}  // namespace

int main(int argc, char** argv) {
  const int F = argc > 1 ? std::atoi(argv[1]) : 64, n = argc > 2 ? std::atoi(argv[2]) : 20000;
  const int rounds = argc > 3 ? std::atoi(argv[3]) : 50, repeats = argc > 4 ? std::atoi(argv[4]) : 5;
  const float K[9] = {180, 0, 0, 0, 180, 0, 320, 240, 1};     // column-major [180 0 320; 0 180 240; 0 0 1]
  vo_ctx* ctx = nullptr;
  CHECK(vo_ctx_create(0, nullptr, &ctx));
  std::vector<float> ref_app, cur_app, ref_pts, cur_pts, model;
  std::vector<int32_t> model_pairs;
  std::vector<vo::Isometry3f> gt;
  for (int f = 0; f < F; ++f) {
    const Pair p = make_pair(n, 4000 + (uint64_t)f, K);
    ref_app.insert(ref_app.end(), p.ref_app.begin(), p.ref_app.end()); cur_app.insert(cur_app.end(), p.cur_app.begin(), p.cur_app.end());
    ref_pts.insert(ref_pts.end(), p.ref_pts.begin(), p.ref_pts.end()); cur_pts.insert(cur_pts.end(), p.cur_pts.begin(), p.cur_pts.end());
    model.insert(model.end(), p.model.begin(), p.model.end()); model_pairs.insert(model_pairs.end(), p.model_pairs.begin(), p.model_pairs.end());
    gt.push_back(p.X_gt);
  }
  vo_frame_batch b{};
  b.n_frames = F; b.n_ref = b.n_cur = b.n_model = b.n_model_pairs = n;
  b.ref_app = upload(ctx, ref_app); b.cur_app = upload(ctx, cur_app);
  b.ref_pts = upload(ctx, ref_pts); b.cur_pts = upload(ctx, cur_pts);
  b.model = upload(ctx, model); b.model_pairs = upload(ctx, model_pairs);
  b.X_prev = nullptr;
  b.rows = 480; b.cols = 640; b.z_near = 0; b.z_far = 10;
  for (int k = 0; k < 9; ++k) b.K[k] = K[k];
  b.kernel_threshold = 10000.f; b.keep_outliers = 0; b.n_iters = rounds; b.radius = 0.1f;
  const size_t Fn = (size_t)F * (size_t)n;
  b.matches = alloc<int32_t>(ctx, 2 * Fn); b.joined = alloc<int32_t>(ctx, 2 * Fn);
  b.model_moved = alloc<float>(ctx, 3 * Fn); b.poses = alloc<float>(ctx, 16 * (size_t)F); b.stats = alloc<float>(ctx, 4 * (size_t)F);
  b.tri_xyz = alloc<float>(ctx, 3 * Fn); b.tri_pairs = alloc<int32_t>(ctx, 2 * Fn); b.tri_app = nullptr;
  b.counts = alloc<int>(ctx, 3 * (size_t)F);
  if (!b.ref_app || !b.cur_app || !b.ref_pts || !b.cur_pts || !b.model || !b.model_pairs || !b.matches || !b.joined ||
      !b.model_moved || !b.poses || !b.stats || !b.tri_xyz || !b.tri_pairs || !b.counts) {
    std::fprintf(stderr, "device allocation failed: %s\n", vo_last_error());
    return 2;
  }
  CHECK(vo_frames_batch_dev(ctx, &b));                        // sizes every workspace
  CHECK(vo_ctx_synchronize(ctx));
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < repeats; ++r) CHECK(vo_frames_batch_dev(ctx, &b));
  CHECK(vo_ctx_synchronize(ctx));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / repeats;
  std::vector<float> poses(16 * (size_t)F), stats(4 * (size_t)F);
  std::vector<int> counts(3 * (size_t)F);
  CHECK(vo_memcpy_d2h(ctx, poses.data(), b.poses, poses.size() * sizeof(float)));
  CHECK(vo_memcpy_d2h(ctx, stats.data(), b.stats, stats.size() * sizeof(float)));
  CHECK(vo_memcpy_d2h(ctx, counts.data(), b.counts, counts.size() * sizeof(int)));
  float worst = 0.f;
  int bad = 0;
  for (int f = 0; f < F; ++f) {
    for (int k = 0; k < 16; ++k) worst = std::max(worst, std::fabs(poses[16 * (size_t)f + k] - gt[(size_t)f].m[k]));
    if (counts[(size_t)f] != n || counts[(size_t)F + f] != n || (int)stats[4 * (size_t)f + 2] != n) ++bad;
  }
  std::printf("batch_frames: %d frames x %d points, %d rounds: %.3f ms per call, %.0f frames/s; worst |T - T_gt| %.2e; "
              "frames with a missing match/join/inlier: %d; triangulated (frame 0): %d\n",
              F, n, rounds, ms, F * 1e3 / ms, worst, bad, counts[2 * (size_t)F]);
  for (void* d : {(void*)b.ref_app, (void*)b.cur_app, (void*)b.ref_pts, (void*)b.cur_pts, (void*)b.model, (void*)b.model_pairs,
                  (void*)b.matches, (void*)b.joined, (void*)b.model_moved, (void*)b.poses, (void*)b.stats, (void*)b.tri_xyz,
                  (void*)b.tri_pairs, (void*)b.counts})
    vo_dev_free(ctx, d);
  vo_ctx_destroy(ctx);
  return (bad == 0 && worst < 2e-3f) ? 0 : 1;
}
