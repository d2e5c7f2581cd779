// batch_frames_mgpu -- BASELINE configs[3] as a NATIVE multi-GPU program: P independent synthetic frame pairs (seeds
// 4000 + p) sharded over the GPUs of one node in contiguous blocks (pair p -> rank floor(p / ceil(P / R)),
// the partition of visual-odometry_amd/dist.py: shard_range restated below), ONE process, one host thread and one vo_ctx
// per device, every rank's share through vo_frames_batch_dev over the C ABI, and at the end of every pass ONE RCCL
// all-gather of the 4x4 poses over xGMI (ncclCommInitAll; each rank thread enqueues its ncclAllGather on its context's
// stream).  There is no data-path collective: the gather is the only exchange.
//   usage: batch_frames_mgpu [gpus=0 (all)] [pairs=1600] [points=50000] [rounds=50] [repeats=3] [frames_per_call=0 (all)]
// VO_MGPU_SHARE_GPU=1 in the environment: a REHEARSAL on a box with one GPU -- `gpus` ranks, every one its own context on device
// 0, the all-gather staged through host memory (RCCL refuses several ranks on one device); its rate is not a scaling number.
// Prints one line per run and one JSON object; exits 0 only when every frame of every rank found all its matches / joins /
// inliers, every pose is the generator's ground truth and every rank holds every other rank's poses after the gather.
// Plain C++ over include/vo_hip.h + rccl.h + the HIP runtime API (device count, nothing else).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "synth.hpp"
#include "vo/shard.hpp"
#include "vo_hip.h"

namespace {

using vo::shard::Agreement;
using vo::shard::Barrier;
using vo::shard::shard_range;      // the partition, the gathered buffer's layout and the go / no-go rule: vo/shard.hpp

struct Pair {
  std::vector<float> ref_app, cur_app, ref_pts, cur_pts, model;
  std::vector<int32_t> model_pairs;
  vo::Isometry3f X_gt;
};

// the frustum-filling pair of SURVEY 8(d) config 2 (as apps/batch_frames.cpp): n landmarks visible in both views of a
// small motion, appearance copied exactly into both images, current image in a random order
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

struct Shared {
  int world, P, n, rounds, repeats, per_call, blk;
  bool share = false;                   // rehearsal: all ranks on device 0, the gather through the host
  std::vector<float> host_poses;        // rehearsal: [world * blk][16]
  std::vector<ncclComm_t> comms;
  Barrier* bar;
  Agreement* agree;                     // per rank: first failure; the common go / no-go before every collective
  std::vector<double> seconds;          // per rank: wall time of the timed passes
  std::vector<float> worst;             // per rank: worst |T - T_gt|
  std::vector<int> bad;                 // per rank: frames with a missing match / join / inlier
  std::vector<int> gather_mismatch;     // per rank: gathered rows that differ from their owner's poses
  std::vector<float> all_poses;         // rank 0's copy of the gathered poses (P rows, global pair order)
};

#define RANK_CHECK(call)                                                                         \
  do {                                                                                           \
    const int rc_ = (call);                                                                      \
    if (rc_ != VO_OK) S.agree->fail(rank, std::string(#call) + ": " + vo_last_error());            \
  } while (0)
#define NCCL_CHECK(call)                                                                         \
  do {                                                                                           \
    const ncclResult_t rc_ = (call);                                                             \
    if (rc_ != ncclSuccess) S.agree->fail(rank, std::string(#call) + ": " + ncclGetErrorString(rc_)); \
  } while (0)

template <class T>
T* dev_alloc(vo_ctx* ctx, size_t n) {
  void* d = nullptr;
  return vo_dev_alloc(ctx, (n ? n : 1) * sizeof(T), &d) == VO_OK ? static_cast<T*>(d) : nullptr;
}

void rank_main(int rank, Shared& S) {
  const float K[9] = {180, 0, 0, 0, 180, 0, 320, 240, 1};     // column-major [180 0 320; 0 180 240; 0 0 1]
  int lo, hi;
  shard_range(S.P, rank, S.world, lo, hi);
  const int F = hi - lo, n = S.n;
  vo_ctx* ctx = nullptr;
  RANK_CHECK(vo_ctx_create(S.share ? 0 : rank, nullptr, &ctx));   // its own stream on device `rank`
  const size_t Fn = (size_t)F * (size_t)n;
  vo_frame_batch b{};
  std::vector<vo::Isometry3f> gt;
  std::vector<void*> owned;
  if (ctx) {
    float* ref_app = dev_alloc<float>(ctx, 10 * Fn); float* cur_app = dev_alloc<float>(ctx, 10 * Fn);
    float* ref_pts = dev_alloc<float>(ctx, 2 * Fn); float* cur_pts = dev_alloc<float>(ctx, 2 * Fn);
    float* model = dev_alloc<float>(ctx, 3 * Fn); int32_t* model_pairs = dev_alloc<int32_t>(ctx, 2 * Fn);
    b.matches = dev_alloc<int32_t>(ctx, 2 * Fn); b.joined = dev_alloc<int32_t>(ctx, 2 * Fn);
    b.model_moved = dev_alloc<float>(ctx, 3 * Fn); b.stats = dev_alloc<float>(ctx, 4 * (size_t)F);
    b.tri_xyz = dev_alloc<float>(ctx, 3 * Fn); b.tri_pairs = dev_alloc<int32_t>(ctx, 2 * Fn); b.tri_app = nullptr;
    b.counts = dev_alloc<int>(ctx, 3 * (size_t)F);
    b.poses = dev_alloc<float>(ctx, 16 * (size_t)S.blk);       // blk rows: the gather needs equal-sized blocks (padding rows stay 0)
    owned = {ref_app, cur_app, ref_pts, cur_pts, model, model_pairs, b.matches, b.joined, b.model_moved, b.stats, b.tri_xyz,
             b.tri_pairs, b.counts, b.poses};
    for (void* d : owned) if (!d) S.agree->fail(rank, std::string("device allocation failed: ") + vo_last_error());
    if (S.agree->ok(rank)) {
      std::vector<float> zero(16 * (size_t)S.blk, 0.f);
      RANK_CHECK(vo_memcpy_h2d(ctx, b.poses, zero.data(), zero.size() * sizeof(float)));
      for (int f = 0; f < F && S.agree->ok(rank); ++f) {       // this rank's pairs: generated here, uploaded, dropped
        const Pair p = make_pair(n, 4000 + (uint64_t)(lo + f), K);
        const size_t at = (size_t)f * (size_t)n;
        RANK_CHECK(vo_memcpy_h2d(ctx, ref_app + 10 * at, p.ref_app.data(), p.ref_app.size() * sizeof(float)));
        RANK_CHECK(vo_memcpy_h2d(ctx, cur_app + 10 * at, p.cur_app.data(), p.cur_app.size() * sizeof(float)));
        RANK_CHECK(vo_memcpy_h2d(ctx, ref_pts + 2 * at, p.ref_pts.data(), p.ref_pts.size() * sizeof(float)));
        RANK_CHECK(vo_memcpy_h2d(ctx, cur_pts + 2 * at, p.cur_pts.data(), p.cur_pts.size() * sizeof(float)));
        RANK_CHECK(vo_memcpy_h2d(ctx, model + 3 * at, p.model.data(), p.model.size() * sizeof(float)));
        RANK_CHECK(vo_memcpy_h2d(ctx, model_pairs + 2 * at, p.model_pairs.data(), p.model_pairs.size() * sizeof(int32_t)));
        gt.push_back(p.X_gt);
      }
    }
    b.ref_app = ref_app; b.cur_app = cur_app; b.ref_pts = ref_pts; b.cur_pts = cur_pts; b.model = model; b.model_pairs = model_pairs;
  }
  b.n_ref = b.n_cur = b.n_model = b.n_model_pairs = n;
  b.X_prev = nullptr;
  b.rows = 480; b.cols = 640; b.z_near = 0; b.z_far = 10;
  for (int k = 0; k < 9; ++k) b.K[k] = K[k];
  b.kernel_threshold = 10000.f; b.keep_outliers = 0; b.n_iters = S.rounds; b.radius = 0.1f;
  float* gathered = ctx ? dev_alloc<float>(ctx, 16 * (size_t)S.blk * (size_t)S.world) : nullptr;
  if (ctx && !gathered) S.agree->fail(rank, "device allocation failed (gather buffer)");

  // one pass = this rank's share in calls of <= per_call frames (every stage one batched launch per call), then -- once
  // EVERY rank has come through its calls (Agreement: nobody enters a collective that another rank will not) -- the gather
  const std::vector<vo::shard::Call> calls = vo::shard::calls_of(F, S.per_call);
  auto pass = [&]() -> bool {
    for (const vo::shard::Call& cl : calls) {
      if (!S.agree->ok(rank)) break;
      vo_frame_batch c = b;
      const int f0 = cl.first, Fc = cl.count;
      const size_t at = (size_t)f0 * (size_t)n;
      c.n_frames = Fc;
      c.ref_app = b.ref_app + 10 * at; c.cur_app = b.cur_app + 10 * at; c.ref_pts = b.ref_pts + 2 * at; c.cur_pts = b.cur_pts + 2 * at;
      c.model = b.model + 3 * at; c.model_pairs = b.model_pairs + 2 * at;
      c.matches = b.matches + 2 * at; c.joined = b.joined + 2 * at; c.model_moved = b.model_moved + 3 * at;
      c.poses = b.poses + 16 * (size_t)f0; c.stats = b.stats + 4 * (size_t)f0;
      c.tri_xyz = b.tri_xyz + 3 * at; c.tri_pairs = b.tri_pairs + 2 * at;
      c.counts = b.counts + 3 * (size_t)f0;                    // this call's [3][Fc] block
      RANK_CHECK(vo_frames_batch_dev(ctx, &c));
    }
    if (!S.agree->all_ok()) return false;
    // the final exchange: SE(3) poses of all ranks (blk x 16 floats each), on the context's stream behind the last launch
    if (S.share) {                                             // by hand: own block to the host, barrier, everybody's blocks back
      RANK_CHECK(vo_memcpy_d2h(ctx, &S.host_poses[16 * (size_t)rank * (size_t)S.blk], b.poses, sizeof(float) * 16 * (size_t)S.blk));
      S.bar->wait();
      RANK_CHECK(vo_memcpy_h2d(ctx, gathered, S.host_poses.data(), sizeof(float) * S.host_poses.size()));
      S.bar->wait();                                           // (the host buffer is rewritten by the next pass)
    } else {
      NCCL_CHECK(ncclAllGather(b.poses, gathered, 16 * (size_t)S.blk, ncclFloat, S.comms[(size_t)rank],
                               reinterpret_cast<hipStream_t>(vo_ctx_stream(ctx))));
    }
    return true;
  };
  // a rank that failed during set-up still meets the others here; set-up failures end the run before the first collective
  if (S.agree->all_ok()) {
    bool good = pass();                                        // sizes every workspace, warms the communicator
    if (good) RANK_CHECK(vo_ctx_synchronize(ctx));
    S.bar->wait();
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < S.repeats && good; ++r) good = pass();
    if (ctx) RANK_CHECK(vo_ctx_synchronize(ctx));
    S.bar->wait();                                             // the job is done when the slowest rank is
    S.seconds[(size_t)rank] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / S.repeats;
  }
  if (S.agree->all_ok()) {
    // checks: own frames against the generator's ground truth, own block of the gathered buffer, everybody's blocks
    std::vector<float> poses(16 * (size_t)std::max(F, 1)), stats(4 * (size_t)std::max(F, 1)), all(16 * (size_t)S.blk * (size_t)S.world);
    std::vector<int> counts(3 * (size_t)std::max(F, 1));
    RANK_CHECK(vo_memcpy_d2h(ctx, poses.data(), b.poses, 16 * (size_t)F * sizeof(float)));
    RANK_CHECK(vo_memcpy_d2h(ctx, stats.data(), b.stats, 4 * (size_t)F * sizeof(float)));
    RANK_CHECK(vo_memcpy_d2h(ctx, counts.data(), b.counts, 3 * (size_t)F * sizeof(int)));
    RANK_CHECK(vo_memcpy_d2h(ctx, all.data(), gathered, all.size() * sizeof(float)));
    float worst = 0.f;
    int bad = 0;
    for (const vo::shard::Call& cl : calls) {
      const int f0 = cl.first, Fc = cl.count;
      for (int f = 0; f < Fc; ++f) {
        const int* c3 = counts.data() + 3 * (size_t)f0;        // [3][Fc]
        for (int k = 0; k < 16; ++k) worst = std::max(worst, std::fabs(poses[16 * (size_t)(f0 + f) + k] - gt[(size_t)(f0 + f)].m[k]));
        if (c3[f] != n || c3[(size_t)Fc + f] != n || (int)stats[4 * (size_t)(f0 + f) + 2] != n) ++bad;
      }
    }
    S.worst[(size_t)rank] = worst; S.bad[(size_t)rank] = bad;
    // own block of the gathered buffer == what this rank computed; every rank's rows are rigid transforms (vo/shard.hpp)
    S.gather_mismatch[(size_t)rank] = vo::shard::own_block_mismatches(all.data(), S.P, S.world, rank, poses.data(), 16) +
                                      vo::shard::rows_not_rigid(all.data(), S.P, S.world);
    if (rank == 0) S.all_poses = vo::shard::to_global_order(all.data(), S.P, S.world, 16);
  }
  if (ctx) {
    for (void* d : owned) if (d) vo_dev_free(ctx, d);
    if (gathered) vo_dev_free(ctx, gathered);
    vo_ctx_destroy(ctx);
  }
}

}  // namespace

int main(int argc, char** argv) {
  int want = argc > 1 ? std::atoi(argv[1]) : 0;
  Shared S;
  S.P = argc > 2 ? std::atoi(argv[2]) : 1600;
  S.n = argc > 3 ? std::atoi(argv[3]) : 50000;
  S.rounds = argc > 4 ? std::atoi(argv[4]) : 50;
  S.repeats = argc > 5 ? std::atoi(argv[5]) : 3;
  S.per_call = argc > 6 ? std::atoi(argv[6]) : 0;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { std::fprintf(stderr, "batch_frames_mgpu: no HIP device (the path has no CPU fallback)\n"); return 2; }
  const char* share_env = std::getenv("VO_MGPU_SHARE_GPU");
  S.share = share_env && share_env[0] == '1';
  if (want <= 0) want = n_dev;
  if (want > n_dev && !S.share) { std::fprintf(stderr, "batch_frames_mgpu: %d GPUs asked for, %d present\n", want, n_dev); return 2; }
  if (S.P < want || S.n < 8 || S.rounds < 0 || S.repeats < 1) { std::fprintf(stderr, "batch_frames_mgpu: need pairs >= gpus, points >= 8, repeats >= 1\n"); return 2; }
  S.world = want;
  S.blk = vo::shard::block_rows(S.P, S.world);                                  // rank 0 holds a largest block
  if (S.share) {
    S.host_poses.assign(16 * (size_t)S.world * (size_t)S.blk, 0.f);
  } else {
    std::vector<int> devs((size_t)S.world);
    std::iota(devs.begin(), devs.end(), 0);
    S.comms.resize((size_t)S.world);
    const ncclResult_t rc = ncclCommInitAll(S.comms.data(), S.world, devs.data());
    if (rc != ncclSuccess) { std::fprintf(stderr, "ncclCommInitAll: %s\n", ncclGetErrorString(rc)); return 2; }
  }
  Barrier bar(S.world);
  Agreement agree(S.world, bar);
  S.bar = &bar; S.agree = &agree;
  S.seconds.assign((size_t)S.world, 0.0); S.worst.assign((size_t)S.world, 0.f);
  S.bad.assign((size_t)S.world, 0); S.gather_mismatch.assign((size_t)S.world, 0);
  std::vector<std::thread> th;
  for (int r = 0; r < S.world; ++r) th.emplace_back(rank_main, r, std::ref(S));
  for (auto& t : th) t.join();
  for (ncclComm_t c : S.comms) ncclCommDestroy(c);
  int fail = 0;
  for (int r = 0; r < S.world; ++r)
    if (!agree.errors()[(size_t)r].empty()) { std::fprintf(stderr, "rank %d: %s\n", r, agree.errors()[(size_t)r].c_str()); fail = 2; }
  if (fail) return fail;
  const double sec = *std::max_element(S.seconds.begin(), S.seconds.end());
  const float worst = *std::max_element(S.worst.begin(), S.worst.end());
  const int bad = std::accumulate(S.bad.begin(), S.bad.end(), 0), mism = std::accumulate(S.gather_mismatch.begin(), S.gather_mismatch.end(), 0);
  std::printf("batch_frames_mgpu: %d GPU(s), %d pairs x %d points, %d rounds, blocks of %d: %.3f ms per pass, %.0f frames/s; "
              "worst |T - T_gt| %.2e; frames with a missing match/join/inlier: %d; gathered rows that differ: %d\n",
              S.world, S.P, S.n, S.rounds, S.blk, sec * 1e3, S.P / sec, worst, bad, mism);
  std::printf("{\"app\": \"batch_frames_mgpu\", \"n_gpus\": %d, \"pairs_total\": %d, \"points\": %d, \"rounds\": %d, \"frames_per_sec\": %.1f, "
              "\"seconds_per_pass\": %.6f, \"scaling\": \"strong\", \"gather\": \"ncclAllGather of %d x 16 floats per rank\", "
              "\"worst_pose_err\": %.3e, \"bad_frames\": %d, \"gather_mismatches\": %d%s}\n",
              S.world, S.P, S.n, S.rounds, S.P / sec, sec, S.blk, worst, bad, mism,
              S.share ? ", \"rehearsal\": \"VO_MGPU_SHARE_GPU=1: all ranks on ONE GPU, gather staged through the host -- not a scaling run\"" : "");
  return (bad == 0 && mism == 0 && worst < 2e-3f) ? 0 : 1;
}
