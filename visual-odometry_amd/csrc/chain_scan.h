// chain_scan.h -- exclusive prefix over the workgroups of ONE launch ("chained scan with decoupled look-back"): what
// lets a stable compaction be a single pass.  The count / scan / scatter triple of rounds 1-4 visited every item twice
// (count pass, writing pass) with a scan launch in between; here a workgroup counts its survivors, publishes the count,
// looks back over its predecessors' published words until it meets one that already knows its inclusive prefix, and
// writes its survivors at that offset -- same ranks, same order, same output bytes.
//
// State per chain (one chain per frame), 8-byte words in device memory, ZEROED by the host before every launch
// (hipMemsetAsync in the launch function -- a memset node under graph capture):
//   word 0      ticket counter: a workgroup's position in the chain is the ticket it draws, not its block index, so
//               every predecessor of a waiting workgroup is already running (no dependence on dispatch order)
//   word 1      set when a look-back gave up (bounded spin: every wave reaches an exit whatever happens)
//   word 2 + b  status of chunk b: (flag << 32) | value; flag 1 = value is the chunk's own count, flag 2 = value is the
//               inclusive prefix up to and including the chunk
// Every access of these words is an 8-byte agent-scope relaxed atomic: flag and value travel in one word, nothing else
// is handed from one workgroup to another, so no fence is needed (per-XCD L2s are not coherent with each other and a
// CU's L1 is never refreshed by other CUs' stores: the atomics go past both).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

namespace vo {

typedef __attribute__((address_space(1))) unsigned long long chain_word;
constexpr int CHAIN_HDR = 2;
constexpr unsigned CHAIN_SPIN_LIMIT = 1u << 22;      // passes over a window before a look-back gives up (~ seconds)

// words of one chain of nb chunks (a multiple of two words: the host zeroes whole 16-byte pieces)
__host__ __device__ inline size_t chain_words(int nb) { return (size_t)((nb + CHAIN_HDR + 1) & ~1); }

__device__ __forceinline__ chain_word* chain_ptr(unsigned long long* p) { return (chain_word*)p; }

// ONE lane: this workgroup's chunk
__device__ __forceinline__ int chain_take_ticket(unsigned long long* st) {
  return (int)__hip_atomic_fetch_add(chain_ptr(st), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int chain_wave_sum(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// All 64 lanes of ONE wave of the workgroup that holds chunk b, whose own count is `total`: publishes it, returns the
// sum of the counts of chunks 0 .. b-1 (the same value in every lane), publishes the inclusive prefix.  -1: gave up
// (word 1 set; the caller must not write -- what it publishes instead keeps its successors inside their arrays).
__device__ __forceinline__ int chain_lookback(unsigned long long* st, int b, int total) {
  chain_word* g = chain_ptr(st) + CHAIN_HDR;
  const int lane = threadIdx.x & 63;
  if (b == 0) {
    if (lane == 0) __hip_atomic_store(g, (2ull << 32) | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
  }
  if (lane == 0) __hip_atomic_store(g + b, (1ull << 32) | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int excl = 0;
  bool failed = false;
  for (int j = b - 1;; j -= 64) {
    const int idx = j - lane;                      // lane 0 looks at the nearest predecessor
    unsigned long long w = 2ull << 32;             // in front of chunk 0: an inclusive prefix of 0
    for (unsigned spins = 0;; ++spins) {
      if (idx >= 0) w = __hip_atomic_load(g + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((w >> 32) != 0)) break;
      if (spins > CHAIN_SPIN_LIMIT) { failed = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (failed) break;
    const unsigned long long incl = __ballot((w >> 32) == 2ull);
    const int first = incl ? __ffsll((long long)incl) - 1 : 64;      // nearest predecessor that knows its inclusive prefix
    excl += chain_wave_sum(lane <= first ? (int)(unsigned)w : 0);
    if (first < 64) break;
  }
  if (lane == 0) {
    if (failed) __hip_atomic_store(chain_ptr(st) + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (a partial sum is <= the true prefix: successors of a chunk that gave up stay inside their arrays)
    __hip_atomic_store(g + b, (2ull << 32) | (unsigned)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return failed ? -1 : excl;
}

}  // namespace vo
