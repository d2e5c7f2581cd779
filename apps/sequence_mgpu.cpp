// sequence_mgpu -- SURVEY 8(e), second row, as a NATIVE multi-GPU program: a real sequence is serial in its pose chain
// (vo_complete.cpp:150-179: frame t needs pose t-1 and its triangulation), but compute_correspondences_images (:156) needs
// the appearances alone.  So the F-1 consecutive frame pairs are dealt to the GPUs of the node in contiguous blocks
// (vo/shard.hpp), every rank -- one host thread and one vo_ctx per device -- matches its block with ONE
// vo_match_appearances_batch_dev call (frames of different sizes), the per-pair counts and the padded pair lists are brought
// together by two ncclAllGather over xGMI, and rank 0 runs the chain on them (vo::DeviceSequence::setMatchesExternal).
// Same outputs as `vo_complete --resident --match-up-front`, bit for bit (tests/test_gpu_multigpu.py).
//   usage: sequence_mgpu <data dir> [output dir] [gpus=0 (all)] [rounds=100] [--exact]
// VO_MGPU_SHARE_GPU=1 in the environment: a REHEARSAL on a box with one GPU -- `gpus` ranks, every one its own context on device
// 0, the two all-gathers staged through host memory (RCCL refuses several ranks on one device): blocks, padding, row mapping and
// the chain on the gathered pairs run exactly as they will on a multi-GPU node; nothing it prints is a scaling number.
// Plain C++ over include/vo/*.hpp + vo_hip.h + rccl.h + the HIP runtime API (device count, nothing else).
#include <cstdio>
#include <iostream>
#include <algorithm>
#include <numeric>
#include <thread>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "vo/shard.hpp"
#include "vo/vo.hpp"

using namespace vo;
using shard::Agreement;
using shard::Barrier;

namespace {

struct Shared {
  int world = 1, P = 0;                  // ranks; consecutive pairs (frames - 1)
  bool share = false;                    // rehearsal: all ranks on device 0, gathers through the host
  std::vector<int> host_counts;          // rehearsal: [world * blk]
  std::vector<int32_t> host_pairs;       //            [world * blk][cap] pairs
  size_t cap = 1;                        // pairs per row: the largest measurement set
  int blk = 0;                           // rows per rank in the gathered buffers
  const std::vector<PointCloudVector<2>>* frames = nullptr;
  std::vector<ncclComm_t> comms;
  Barrier* bar = nullptr;
  Agreement* agree = nullptr;
  // rank 0's gathered buffers (device memory of the default context): [world * blk] counts, [world * blk][cap] pairs
  int* g_counts = nullptr;
  int32_t* g_pairs = nullptr;
  std::vector<int> pairs_matched;        // per rank: pairs found in its block (for the report)
};

template <class T>
T* dev_alloc(vo_ctx* ctx, size_t n) {
  void* d = nullptr;
  return vo_dev_alloc(ctx, (n ? n : 1) * sizeof(T), &d) == VO_OK ? static_cast<T*>(d) : nullptr;
}

#define RANK_CHECK(call)                                                                          \
  do {                                                                                            \
    const int rc_ = (call);                                                                       \
    if (rc_ != VO_OK) S.agree->fail(rank, std::string(#call) + ": " + vo_last_error());           \
  } while (0)
#define NCCL_CHECK(call)                                                                          \
  do {                                                                                            \
    const ncclResult_t rc_ = (call);                                                              \
    if (rc_ != ncclSuccess) S.agree->fail(rank, std::string(#call) + ": " + ncclGetErrorString(rc_)); \
  } while (0)

// one rank: match the pairs (k, k+1), k in [lo, hi), of its block; all-gather counts and pairs
void rank_main(int rank, Shared& S) {
  int lo, hi;
  shard::shard_range(S.P, rank, S.world, lo, hi);
  const int np = hi - lo;
  const size_t cap = S.cap;
  vo_ctx* own = nullptr;
  vo_ctx* ctx = nullptr;
  if (rank == 0) ctx = default_context().handle();           // the chain runs on this context: same stream, same memory
  else { RANK_CHECK(vo_ctx_create(S.share ? 0 : rank, nullptr, &own)); ctx = own; }
  float* d_app = nullptr; int* d_n = nullptr; int32_t* d_pairs = nullptr; int* d_cnt = nullptr;
  int32_t* g_pairs = nullptr; int* g_counts = nullptr;
  if (ctx) {
    // the block's frames lo .. hi (np + 1 of them) as [np + 1][cap][10]: pair k's two sets are rows k - lo and k - lo + 1
    const int nf = np > 0 ? np + 1 : 0;
    std::vector<float> app(10 * cap * (size_t)std::max(nf, 1), 0.f);
    std::vector<int> sizes((size_t)std::max(nf, 1), 0);
    for (int t = 0; t < nf; ++t) {
      const auto& f = (*S.frames)[(size_t)(lo + t)];
      sizes[(size_t)t] = (int)f.size();
      if (f.size()) std::memcpy(&app[10 * cap * (size_t)t], f.appearances()[0].data(), sizeof(float) * 10 * f.size());
    }
    d_app = dev_alloc<float>(ctx, app.size()); d_n = dev_alloc<int>(ctx, sizes.size());
    d_pairs = dev_alloc<int32_t>(ctx, 2 * cap * (size_t)S.blk); d_cnt = dev_alloc<int>(ctx, (size_t)S.blk);
    g_pairs = dev_alloc<int32_t>(ctx, 2 * cap * (size_t)S.blk * (size_t)S.world); g_counts = dev_alloc<int>(ctx, (size_t)S.blk * (size_t)S.world);
    if (!d_app || !d_n || !d_pairs || !d_cnt || !g_pairs || !g_counts) S.agree->fail(rank, std::string("device allocation failed: ") + vo_last_error());
    if (S.agree->ok(rank)) {
      const std::vector<int> zeros((size_t)S.blk, 0);        // padding rows (behind a shorter block) count zero pairs
      RANK_CHECK(vo_memcpy_h2d(ctx, d_cnt, zeros.data(), zeros.size() * sizeof(int)));
      RANK_CHECK(vo_memcpy_h2d(ctx, d_app, app.data(), app.size() * sizeof(float)));
      RANK_CHECK(vo_memcpy_h2d(ctx, d_n, sizes.data(), sizes.size() * sizeof(int)));
      // the frame is a grid dimension of the batched matcher: at most 65535 pairs per call (as DeviceSequence::run does)
      for (int p0 = 0; p0 < np && S.agree->ok(rank); p0 += 65535) {
        const int k = std::min(65535, np - p0);
        RANK_CHECK(vo_match_appearances_batch_dev(ctx, k, d_app + 10 * cap * (size_t)p0, (int)cap, d_n + p0, d_app + 10 * cap * (size_t)(p0 + 1), (int)cap,
                                                  d_n + p0 + 1, 0.1f, d_pairs + 2 * cap * (size_t)p0, d_cnt + p0));
      }
    }
  }
  if (S.agree->all_ok()) {                                   // nobody enters the collectives unless everybody does
    if (S.share) {
      // the all-gathers by hand: own block to the host, barrier, everybody's blocks back (same layout: rank r at r * blk)
      RANK_CHECK(vo_memcpy_d2h(ctx, &S.host_counts[(size_t)rank * (size_t)S.blk], d_cnt, sizeof(int) * (size_t)S.blk));
      RANK_CHECK(vo_memcpy_d2h(ctx, &S.host_pairs[2 * cap * (size_t)rank * (size_t)S.blk], d_pairs, sizeof(int32_t) * 2 * cap * (size_t)S.blk));
      S.bar->wait();
      RANK_CHECK(vo_memcpy_h2d(ctx, g_counts, S.host_counts.data(), sizeof(int) * S.host_counts.size()));
      RANK_CHECK(vo_memcpy_h2d(ctx, g_pairs, S.host_pairs.data(), sizeof(int32_t) * S.host_pairs.size()));
    } else {
      hipStream_t st = reinterpret_cast<hipStream_t>(vo_ctx_stream(ctx));
      NCCL_CHECK(ncclGroupStart());
      NCCL_CHECK(ncclAllGather(d_cnt, g_counts, (size_t)S.blk, ncclInt32, S.comms[(size_t)rank], st));
      NCCL_CHECK(ncclAllGather(d_pairs, g_pairs, 2 * cap * (size_t)S.blk, ncclInt32, S.comms[(size_t)rank], st));
      NCCL_CHECK(ncclGroupEnd());
      // a rank whose part of the collective failed takes its communicator down, so that the others -- who may already sit
      // inside the all-gather -- come back with an error instead of waiting for it forever
      if (!S.agree->ok(rank)) { (void)ncclCommAbort(S.comms[(size_t)rank]); S.comms[(size_t)rank] = nullptr; }   // (abort frees it)
    }
    RANK_CHECK(vo_ctx_synchronize(ctx));
    std::vector<int> cnt((size_t)S.blk, 0);
    RANK_CHECK(vo_memcpy_d2h(ctx, cnt.data(), d_cnt, cnt.size() * sizeof(int)));
    S.pairs_matched[(size_t)rank] = std::accumulate(cnt.begin(), cnt.begin() + np, 0);
    if (rank == 0) { S.g_counts = g_counts; S.g_pairs = g_pairs; g_counts = nullptr; g_pairs = nullptr; }   // kept for the chain
  }
  S.bar->wait();
  if (ctx) for (void* d : {(void*)d_app, (void*)d_n, (void*)d_pairs, (void*)d_cnt, (void*)g_pairs, (void*)g_counts}) if (d) vo_dev_free(ctx, d);
  if (own) vo_ctx_destroy(own);
}

void write_poses_raw(const std::string& file, const IsometryVector& trajectory) {
  std::FILE* f = std::fopen(file.c_str(), "w");
  if (!f) return;
  for (const auto& X : trajectory) {
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) std::fprintf(f, "%.9g ", X(r, c));
    std::fprintf(f, "\n");
  }
  std::fclose(f);
}

}  // namespace

int main(int argc, char* argv[]) {
  bool exact = false;
  std::vector<std::string> pos;
  for (int i = 1; i < argc; ++i) {
    const std::string a(argv[i]);
    if (a == "--exact") exact = true;
    else if (a.rfind("--", 0) == 0) { std::cout << "unknown option " << a << std::endl; return -1; }
    else pos.push_back(a);
  }
  if (pos.empty()) { std::cout << "usage: sequence_mgpu <data dir> [output dir] [gpus=0 (all)] [rounds=100] [--exact]" << std::endl; return -1; }
  std::string path(pos[0]);
  if (path.back() != '/') path.push_back('/');
  std::string out = pos.size() > 1 ? pos[1] : ".";
  if (out.back() != '/') out.push_back('/');
  int want = pos.size() > 2 ? std::atoi(pos[2].c_str()) : 0;
  const int rounds = pos.size() > 3 ? std::atoi(pos[3].c_str()) : 100;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { std::fprintf(stderr, "sequence_mgpu: no HIP device (the path has no CPU fallback)\n"); return 2; }
  const char* share_env = std::getenv("VO_MGPU_SHARE_GPU");
  const bool share = share_env && share_env[0] == '1';
  if (want <= 0) want = n_dev;
  if (want > n_dev && !share) { std::fprintf(stderr, "sequence_mgpu: %d GPUs asked for, %d present\n", want, n_dev); return 2; }
  try {
    save_gt_trajectory(path + "trajectory.dat", out + "trajectory_gt.txt");
    const std::regex pattern("^meas-\\d.*\\.dat$");
    std::set<std::string> files;
    if (!get_file_names(path, files, pattern)) { std::cout << "unable to open directory\n"; return -1; }
    if (files.size() < 2) { std::cout << "need at least two measurement files\n"; return -1; }
    std::vector<PointCloudVector<2>> frames;
    std::vector<std::string> names(files.begin(), files.end());
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

    Shared S;
    S.world = want; S.P = (int)frames.size() - 1; S.frames = &frames; S.share = share;
    for (const auto& f : frames) S.cap = std::max(S.cap, f.size());
    S.blk = shard::block_rows(S.P, S.world);
    S.pairs_matched.assign((size_t)S.world, 0);
    if (share) {
      S.host_counts.assign((size_t)S.world * (size_t)S.blk, 0);
      S.host_pairs.assign(2 * S.cap * (size_t)S.world * (size_t)S.blk, 0);
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
    DeviceSequence seq(cam, frames, rounds);                 // (creates the default context on device 0 before the ranks start)
    if (seq.capacity() != S.cap) { std::fprintf(stderr, "sequence_mgpu: capacity mismatch\n"); return 2; }
    std::vector<std::thread> th;
    for (int r = 0; r < S.world; ++r) th.emplace_back(rank_main, r, std::ref(S));
    for (auto& t : th) t.join();
    for (ncclComm_t c : S.comms) if (c) ncclCommDestroy(c);
    int fail = 0;
    for (int r = 0; r < S.world; ++r)
      if (!agree.errors()[(size_t)r].empty()) { std::fprintf(stderr, "rank %d: %s\n", r, agree.errors()[(size_t)r].c_str()); fail = 2; }
    if (fail) return fail;

    // rank 0: the chain on the gathered pairs -- pair (t-1, t) is item t-1, found in its owner's block
    std::vector<int> row_of((size_t)S.P);
    for (int p = 0; p < S.P; ++p) row_of[(size_t)p] = (int)shard::gathered_row(S.P, S.world, p);
    seq.setExact(exact);
    seq.setMatchesExternal(S.g_pairs, S.g_counts, row_of);
    seq.run();
    const IsometryVector trajectory = seq.trajectory();      // waits for the chain
    PointCloudVector<3> map;
    map.update(seq.cloud(1));
    Isometry3f history = trajectory[1].inverse();
    long total_matches = 0;
    for (int t = 1; t < seq.frames(); ++t) {
      int n_match, n_join, n_tri;
      seq.counts(t, n_match, n_join, n_tri);
      total_matches += n_match;
      if (t < 2) continue;
      const Isometry3f& X = trajectory[(size_t)t];
      std::printf("%s: %d matches, %d model correspondences, t = % .5f % .5f % .5f\n", names[(size_t)t].c_str(), n_match, n_join,
                  X(0, 3), X(1, 3), X(2, 3));
      map.update(history * seq.cloud(t));
      history = history * X.inverse();
    }
    map = H * map;
    write_eigen_vectors_to_file(out + "map.txt", map.points());
    write_eigen_vectors_to_file(out + "map_appearances.txt", map.appearances());
    save_trajectory(out + "trajectory_est_complete.txt", trajectory, H);
    save_trajectory(out + "trajectory_est_data.txt", trajectory, H, true);
    write_poses_raw(out + "poses_raw.txt", trajectory);
    const long by_ranks = std::accumulate(S.pairs_matched.begin(), S.pairs_matched.end(), 0L);
    std::printf("{\"app\": \"sequence_mgpu\", \"n_gpus\": %d, \"frames\": %d, \"pairs_per_rank\": %d, \"matches_total\": %ld, "
                "\"matches_found_by_the_ranks\": %ld, \"gather\": \"%s (counts, %zu padded pairs per row)\"%s}\n",
                S.world, seq.frames(), S.blk, total_matches, by_ranks, share ? "staged through the host" : "2 x ncclAllGather", S.cap,
                share ? ", \"rehearsal\": \"VO_MGPU_SHARE_GPU=1: all ranks on ONE GPU -- not a scaling run\"" : "");
    vo_dev_free(default_context().handle(), S.g_pairs);
    vo_dev_free(default_context().handle(), S.g_counts);
    return total_matches == by_ranks ? 0 : 1;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "sequence_mgpu: %s\n", e.what());
    return 2;
  }
}
