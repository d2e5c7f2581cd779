// hop_latency.hip -- what it costs to hand 32 values from one workgroup to another INSIDE a launch (DESIGN.md section 4.8:
// would helper workgroups on the 56 CUs that a 200-problem call leaves idle pay?).  Workgroup 0 and workgroup `peer` of a
// 256-workgroup launch bounce a row of 32 tagged 8-byte words (tag << 32 | payload, agent-scope relaxed atomics: the only
// accesses that go past a CU's L1 and an XCD's L2) back and forth; every other workgroup exits at once.  Consecutive
// workgroups go to consecutive XCDs, so peer = 8 stays on workgroup 0's XCD and peer = 1 .. 7 crosses to another.
//   build: hipcc --offload-arch=gfx950 -O2 -o bin/hop_latency hop_latency.hip      usage: hop_latency [round_trips=2000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(1))) unsigned long long gword;

__global__ __launch_bounds__(64) void pingpong(unsigned long long* a, unsigned long long* b, int peer, int trips, long long* ticks, int* gave_up) {
  const int lane = threadIdx.x;
  if (blockIdx.x != 0 && (int)blockIdx.x != peer) return;
  const bool first = blockIdx.x == 0;
  gword* mine = (gword*)(first ? a : b) + (lane & 31);      // what I write
  gword* theirs = (gword*)(first ? b : a) + (lane & 31);    // what I wait for
  const long long t0 = wall_clock64();
  for (int r = 1; r <= trips; ++r) {
    if (first && lane < 32) __hip_atomic_store(mine, ((unsigned long long)r << 32) | (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long w = 0;
    unsigned spins = 0;
    for (;; ++spins) {
      w = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((int)(w >> 32) >= r)) break;
      if (spins > (1u << 22)) { if (lane == 0) *gave_up = r; return; }
    }
    if (!first && lane < 32) __hip_atomic_store(mine, ((unsigned long long)r << 32) | (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (first && lane == 0) *ticks = wall_clock64() - t0;
}

int main(int argc, char** argv) {
  const int trips = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned long long *a, *b;
  long long* ticks;
  int* gave_up;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&ticks, 8); hipMalloc(&gave_up, 4);
  int rate_khz = 0;
  hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  if (rate_khz <= 0) rate_khz = 100000;
  std::printf("{\"round_trips\": %d, \"wall_clock_khz\": %d, \"round_trip_us\": {", trips, rate_khz);
  const int peers[] = {8, 16, 1, 2, 3, 5, 7, 9, 201, 255};
  for (int k = 0; k < (int)(sizeof(peers) / sizeof(peers[0])); ++k) {
    double best = 1e30;
    int failed = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(a, 0, 256); hipMemset(b, 0, 256); hipMemset(ticks, 0, 8); hipMemset(gave_up, 0, 4);
      hipLaunchKernelGGL(pingpong, dim3(256), dim3(64), 0, 0, a, b, peers[k], trips, ticks, gave_up);
      long long t = 0;
      hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      hipMemcpy(&failed, gave_up, 4, hipMemcpyDeviceToHost);
      const double us = (double)t / rate_khz * 1e3 / trips;
      if (!failed && us < best) best = us;
    }
    std::printf("%s\"%d\": %.3f", k ? ", " : "", peers[k], failed ? -1.0 : best);
  }
  std::printf("}}\n");
  return 0;
}
