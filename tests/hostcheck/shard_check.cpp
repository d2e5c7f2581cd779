// shard_check -- the rank bookkeeping of the native multi-GPU drivers (include/vo/shard.hpp) without a GPU: every "rank" is
// a host thread with fake pose buffers, the all-gather is a memcpy between them behind a barrier.  Worlds of 2, 3 and 8
// ranks, uneven item counts (13, 1601), items < ranks, per-call slicing, own-block / foreign-block checks, global order,
// and the rule that no rank enters a collective alone when another one has failed.
#include <atomic>
#include <cstdio>
#include <thread>

#include "vo/shard.hpp"

using namespace vo::shard;

static void pose_of(int p, float* T) {                      // a "pose" that names its item: column-major 4x4, last row 0 0 0 1
  for (int k = 0; k < 16; ++k) T[k] = (float)(1000 * p + k);
  T[3] = T[7] = T[11] = 0.f; T[15] = 1.f;
}

struct Run {
  int P, world, per_call, fail_rank;                        // fail_rank: -1 none; >= 0 fails at set-up; <= -2: rank (-2 - x) fails after the first pass
  std::vector<std::vector<float>> gathered;                 // per rank
  std::vector<std::vector<float>> block;                    // per rank: blk x 16
  std::atomic<int> collectives{0};
  int failures = 0;
};

static void rank_main(int rank, Run& R, Barrier& bar, Agreement& agree, std::vector<int>& bad) {
  const int blk = block_rows(R.P, R.world);
  int lo, hi;
  shard_range(R.P, rank, R.world, lo, hi);
  const int F = hi - lo;
  R.block[(size_t)rank].assign(16 * (size_t)blk, 0.f);
  R.gathered[(size_t)rank].assign(16 * (size_t)blk * (size_t)R.world, -1.f);
  if (R.fail_rank == rank) agree.fail(rank, "set-up failed (injected)");
  if (!agree.all_ok()) return;                              // everybody leaves before the first collective
  for (int pass = 0; pass < 2; ++pass) {
    int covered = 0;
    for (const Call& c : calls_of(F, R.per_call)) {         // this rank's share, call by call, rows block-local
      if (c.first != covered) ++bad[(size_t)rank];
      for (int f = 0; f < c.count; ++f) pose_of(lo + c.first + f, &R.block[(size_t)rank][16 * (size_t)(c.first + f)]);
      covered += c.count;
    }
    if (covered != F) ++bad[(size_t)rank];
    if (pass == 1 && R.fail_rank <= -2 && rank == -2 - R.fail_rank) agree.fail(rank, "second pass failed (injected)");
    if (!agree.all_ok()) return;                            // agreed BEFORE the collective: nobody waits in it alone
    ++R.collectives;
    // the all-gather: every rank's block into every rank's buffer at rank * blk
    bar.wait();
    for (int r = 0; r < R.world; ++r)
      std::memcpy(&R.gathered[(size_t)rank][16 * (size_t)r * (size_t)blk], R.block[(size_t)r].data(), sizeof(float) * 16 * (size_t)blk);
    bar.wait();
  }
  const float* g = R.gathered[(size_t)rank].data();
  bad[(size_t)rank] += own_block_mismatches(g, R.P, R.world, rank, R.block[(size_t)rank].data(), 16);
  bad[(size_t)rank] += rows_not_rigid(g, R.P, R.world);
  const std::vector<float> global = to_global_order(g, R.P, R.world, 16);
  for (int p = 0; p < R.P; ++p) {
    float T[16];
    pose_of(p, T);
    if (std::memcmp(&global[16 * (size_t)p], T, sizeof(T)) != 0) ++bad[(size_t)rank];
    if (std::memcmp(g + 16 * gathered_row(R.P, R.world, p), T, sizeof(T)) != 0) ++bad[(size_t)rank];
    const int o = owner_of(R.P, R.world, p);
    int l2, h2;
    shard_range(R.P, o, R.world, l2, h2);
    if (!(l2 <= p && p < h2)) ++bad[(size_t)rank];
  }
  // padding rows (behind a shorter block) stay what the owner left there: zeros
  for (int r = 0; r < R.world; ++r) {
    int l2, h2;
    shard_range(R.P, r, R.world, l2, h2);
    for (int f = h2 - l2; f < blk; ++f)
      for (int k = 0; k < 16; ++k) if (g[16 * ((size_t)r * blk + f) + k] != 0.f) ++bad[(size_t)rank];
  }
}

static int run(int P, int world, int per_call, int fail_rank) {
  Run R;
  R.P = P; R.world = world; R.per_call = per_call; R.fail_rank = fail_rank;
  R.gathered.resize((size_t)world); R.block.resize((size_t)world);
  Barrier bar(world);
  Agreement agree(world, bar);
  std::vector<int> bad((size_t)world, 0);
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r) th.emplace_back(rank_main, r, std::ref(R), std::ref(bar), std::ref(agree), std::ref(bad));
  for (auto& t : th) t.join();                              // (a rank left alone in a collective would hang here)
  int total_bad = 0;
  for (int b : bad) total_bad += b;
  int n_err = 0;
  for (const std::string& e : agree.errors()) n_err += e.empty() ? 0 : 1;
  const int expect_collectives = fail_rank >= 0 ? 0 : (fail_rank <= -2 ? world : 2 * world);
  const bool ok = total_bad == 0 && R.collectives == expect_collectives && n_err == (fail_rank == -1 ? 0 : 1);
  std::printf("P %d world %d per_call %d fail %d: blocks of %d, collectives entered %d (expected %d), errors %d, bad %d -> %s\n", P, world,
              per_call, fail_rank, block_rows(P, world), (int)R.collectives, expect_collectives, n_err, total_bad, ok ? "ok" : "FAILED");
  return ok ? 0 : 1;
}

int main() {
  int fails = 0;
  // partition basics
  for (int world : {1, 2, 3, 8})
    for (int P : {1, 5, 8, 13, 1600, 1601}) {
      int covered = 0, prev_hi = 0, biggest = 0;
      for (int r = 0; r < world; ++r) {
        int lo, hi;
        shard_range(P, r, world, lo, hi);
        if (lo != prev_hi || hi < lo) ++fails;
        prev_hi = hi; covered += hi - lo; biggest = std::max(biggest, hi - lo);
      }
      if (covered != P || prev_hi != P || biggest != block_rows(P, world)) { ++fails; std::printf("partition P %d world %d FAILED\n", P, world); }
    }
  for (int world : {2, 3, 8})
    for (int P : {13, 1601})
      for (int per_call : {0, 1, 4, 10, 200}) fails += run(P, world, per_call, -1);
  fails += run(5, 8, 0, -1);                                // fewer items than ranks: three ranks hold nothing
  fails += run(8, 8, 3, -1);
  // failure propagation: a rank that fails at set-up, a rank that fails between the passes -- every thread must come back
  for (int world : {2, 3, 8}) {
    fails += run(13, world, 4, 0);
    fails += run(13, world, 4, world - 1);
    fails += run(1601, world, 10, -2);                      // rank 0 fails in the second pass
    fails += run(1601, world, 10, -2 - (world - 1));        // the last rank does
  }
  std::printf(fails ? "shard_check: %d FAILED\n" : "shard_check: all ok\n", fails);
  return fails ? 1 : 0;
}
