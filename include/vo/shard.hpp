// vo/shard.hpp -- the rank bookkeeping of the native multi-GPU drivers (apps/batch_frames_mgpu.cpp, apps/sequence_mgpu.cpp),
// host-only and free of any GPU call, so that tests/hostcheck/shard_check.cpp can drive it with fake buffers at any world
// size.  SURVEY 8(e): items (independent frame pairs, or the consecutive pairs (t, t+1) of a sequence) are dealt to the
// ranks in contiguous blocks, every rank works on its block alone, and ONE all-gather of equal-sized padded blocks brings
// the results together (ncclAllGather wants equal counts: a block is padded to the largest one's size).  The same
// partition as visual-odometry_amd/dist.py (shard_range / StrongPlan), which the torch.distributed route uses.
#pragma once

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace vo {
namespace shard {

//! contiguous blocks that differ by at most one item: items [lo, hi) belong to `rank`
inline void shard_range(int n_items, int rank, int world, int& lo, int& hi) {
  const int base = n_items / world, rem = n_items % world;
  lo = rank * base + std::min(rank, rem);
  hi = lo + base + (rank < rem ? 1 : 0);
}
//! rows of every rank's block in the gathered buffer (= the largest block: rank 0's)
inline int block_rows(int n_items, int world) { return (n_items + world - 1) / world; }
//! the rank that owns item p
inline int owner_of(int n_items, int world, int p) {
  const int base = n_items / world, rem = n_items % world;
  const int big = rem * (base + 1);                 // items held by the ranks with one item more
  return p < big ? p / (base + 1) : rem + (base ? (p - big) / base : 0);
}
//! row of item p in the gathered buffer: its owner's block starts at owner * blk
inline size_t gathered_row(int n_items, int world, int p) {
  const int r = owner_of(n_items, world, p);
  int lo, hi;
  shard_range(n_items, r, world, lo, hi);
  return (size_t)r * (size_t)block_rows(n_items, world) + (size_t)(p - lo);
}

//! a rank's share in calls of at most per_call items (per_call <= 0: one call)
struct Call { int first, count; };
inline std::vector<Call> calls_of(int n_local, int per_call) {
  std::vector<Call> c;
  const int step = per_call > 0 ? per_call : std::max(n_local, 1);
  for (int f0 = 0; f0 < n_local; f0 += step) c.push_back({f0, std::min(step, n_local - f0)});
  return c;
}

//! gathered blocks (world x blk rows of `width` elements, padding rows behind the shorter blocks) -> items in global order
template <class T>
std::vector<T> to_global_order(const T* gathered, int n_items, int world, int width) {
  const int blk = block_rows(n_items, world);
  std::vector<T> out((size_t)n_items * (size_t)width);
  for (int r = 0; r < world; ++r) {
    int lo, hi;
    shard_range(n_items, r, world, lo, hi);
    if (hi > lo) std::memcpy(&out[(size_t)lo * width], gathered + (size_t)r * blk * width, sizeof(T) * (size_t)(hi - lo) * width);
  }
  return out;
}
//! rows of this rank's own block of the gathered buffer that differ from what the rank computed
template <class T>
int own_block_mismatches(const T* gathered, int n_items, int world, int rank, const T* own, int width) {
  int lo, hi;
  shard_range(n_items, rank, world, lo, hi);
  const int blk = block_rows(n_items, world);
  int bad = 0;
  for (int f = 0; f < hi - lo; ++f)
    if (std::memcmp(gathered + ((size_t)rank * blk + f) * width, own + (size_t)f * width, sizeof(T) * width) != 0) ++bad;
  return bad;
}
//! live rows (any rank's) of gathered 4x4 column-major poses whose last row is not (0 0 0 1): a rigid transform has it
inline int rows_not_rigid(const float* gathered, int n_items, int world) {
  const int blk = block_rows(n_items, world);
  int bad = 0;
  for (int r = 0; r < world; ++r) {
    int lo, hi;
    shard_range(n_items, r, world, lo, hi);
    for (int f = 0; f < hi - lo; ++f) {
      const float* T = gathered + 16 * ((size_t)r * blk + f);
      if (!(T[3] == 0.f && T[7] == 0.f && T[11] == 0.f && T[15] == 1.f)) ++bad;
    }
  }
  return bad;
}

//! the ranks of one process are host threads: a reusable barrier
class Barrier {
 public:
  explicit Barrier(int n) : n_(n) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m_);
    const int ph = phase_;
    if (++waiting_ == n_) { waiting_ = 0; ++phase_; cv_.notify_all(); }
    else cv_.wait(lk, [&] { return phase_ != ph; });
  }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  int n_, waiting_ = 0, phase_ = 0;
};

//! One error slot per rank and the rule that keeps a collective from being entered by some ranks only: every rank reports
//! its set-up (or step) outcome, all meet at the barrier, and ALL take the same decision from the same slots -- go on only
//! when no rank has failed.  A rank that failed still meets the others here, so nobody waits alone.
class Agreement {
 public:
  Agreement(int world, Barrier& bar) : err_((size_t)world), bar_(bar) {}
  void fail(int rank, const std::string& what) { if (err_[(size_t)rank].empty()) err_[(size_t)rank] = what; }   // (a rank writes its own slot only)
  bool ok(int rank) const { return err_[(size_t)rank].empty(); }
  //! barrier, then the common verdict
  bool all_ok() {
    bar_.wait();
    bool good = true;
    for (const std::string& e : err_) good = good && e.empty();
    bar_.wait();                                      // nobody may fail() again before everyone has read the slots
    return good;
  }
  const std::vector<std::string>& errors() const { return err_; }
 private:
  std::vector<std::string> err_;
  Barrier& bar_;
};

}  // namespace shard
}  // namespace vo
