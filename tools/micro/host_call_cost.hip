// host_call_cost.hip -- what a vo_picp_one_round call is made of on the host:
//   * the full comparison of 50 000 index pairs (400 KB) with their host copy: glibc memcmp and bytes_equal,
//     both arrays warm (nothing else touches the cache between two calls of the loop);
//   * one kernel launch on a stream that is not waited for (hipLaunchKernelGGL of a kernel with the round kernel's
//     argument list), in bursts of 50 as the loop issues them.
//   build: hipcc --offload-arch=gfx950 -O2 -o bin/host_call_cost host_call_cost.hip      usage: host_call_cost [pairs=50000]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <immintrin.h>
#include <string.h>

// the alternative measured against glibc memcmp: 512-byte blocks, four independent OR chains, one test per block
__attribute__((target("avx2"))) static inline bool bytes_equal(const void* a, const void* b, size_t n) {
  const char* p = static_cast<const char*>(a);
  const char* q = static_cast<const char*>(b);
  size_t i = 0;
  for (; i + 512 <= n; i += 512) {
    __m256i acc0 = _mm256_setzero_si256(), acc1 = acc0, acc2 = acc0, acc3 = acc0;
#pragma GCC unroll 4
    for (int k = 0; k < 512; k += 128) {
      acc0 = _mm256_or_si256(acc0, _mm256_xor_si256(_mm256_loadu_si256((const __m256i*)(p + i + k)), _mm256_loadu_si256((const __m256i*)(q + i + k))));
      acc1 = _mm256_or_si256(acc1, _mm256_xor_si256(_mm256_loadu_si256((const __m256i*)(p + i + k + 32)), _mm256_loadu_si256((const __m256i*)(q + i + k + 32))));
      acc2 = _mm256_or_si256(acc2, _mm256_xor_si256(_mm256_loadu_si256((const __m256i*)(p + i + k + 64)), _mm256_loadu_si256((const __m256i*)(q + i + k + 64))));
      acc3 = _mm256_or_si256(acc3, _mm256_xor_si256(_mm256_loadu_si256((const __m256i*)(p + i + k + 96)), _mm256_loadu_si256((const __m256i*)(q + i + k + 96))));
    }
    const __m256i acc = _mm256_or_si256(_mm256_or_si256(acc0, acc1), _mm256_or_si256(acc2, acc3));
    if (!_mm256_testz_si256(acc, acc)) return false;
  }
  return i == n || memcmp(p + i, q + i, n - i) == 0;
}



using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

__global__ void probe_kernel(const void* P, void* S, float* pk, size_t cap, float* partials, int it, int nb) {
  if (it < 0) partials[0] = (float)nb + (float)cap + (pk != nullptr) + (P != nullptr) + (S != nullptr);
}

int main(int argc, char** argv) {
  const size_t pairs = argc > 1 ? (size_t)atoll(argv[1]) : 50000;
  const size_t bytes = pairs * 8;
  std::vector<int32_t> a(2 * pairs), b(2 * pairs);
  for (size_t i = 0; i < 2 * pairs; ++i) a[i] = b[i] = (int32_t)(i * 2654435761u);
  volatile int sink = 0;
  const int reps = 2000;
  for (int w = 0; w < 2; ++w) {
    auto t0 = clk::now();
    for (int r = 0; r < reps; ++r) { sink = sink + (memcmp(a.data(), b.data(), bytes) == 0); asm volatile("" ::: "memory"); }
    auto t1 = clk::now();
    for (int r = 0; r < reps; ++r) { sink = sink + bytes_equal(a.data(), b.data(), bytes); asm volatile("" ::: "memory"); }
    auto t2 = clk::now();
    if (w) std::printf("{\"pairs\": %zu, \"memcmp_us\": %.3f, \"bytes_equal_us\": %.3f, ", pairs, us(t0, t1) / reps, us(t1, t2) / reps);
  }
  hipStream_t st;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { std::printf("\"launch_us\": null}\n"); return 1; }
  float* d = nullptr;
  (void)hipMalloc(&d, 1024);
  double launch = 0, launch_cmp = 0;
  for (int w = 0; w < 3; ++w) {
    launch = launch_cmp = 0;
    for (int burst = 0; burst < 100; ++burst) {
      auto t0 = clk::now();
      for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(probe_kernel, dim3(196), dim3(256), 0, st, (const void*)d, (void*)d, d, (size_t)50000, d, i, 196);
      auto t1 = clk::now();
      (void)hipStreamSynchronize(st);
      launch += us(t0, t1);
      // the same with the comparison between two launches (what the loop does)
      t0 = clk::now();
      for (int i = 0; i < 50; ++i) {
        hipLaunchKernelGGL(probe_kernel, dim3(196), dim3(256), 0, st, (const void*)d, (void*)d, d, (size_t)50000, d, i, 196);
        sink = sink + bytes_equal(a.data(), b.data(), bytes);
      }
      t1 = clk::now();
      (void)hipStreamSynchronize(st);
      launch_cmp += us(t0, t1);
    }
  }
  std::printf("\"launch_us\": %.3f, \"launch_plus_compare_us\": %.3f", launch / 5000, launch_cmp / 5000);
  // one hipGraphLaunch of a captured run of k such kernels (what enqueueing k rounds ahead of the caller would cost per call)
  for (int k : {1, 2, 4, 8}) {
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) break;
    for (int i = 0; i < k; ++i) hipLaunchKernelGGL(probe_kernel, dim3(196), dim3(256), 0, st, (const void*)d, (void*)d, d, (size_t)50000, d, i, 196);
    if (hipStreamEndCapture(st, &g) != hipSuccess || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) break;
    double t = 0;
    for (int w = 0; w < 3; ++w) {
      t = 0;
      for (int burst = 0; burst < 100; ++burst) {
        auto t0 = clk::now();
        for (int i = 0; i < 50 / k + 1; ++i) (void)hipGraphLaunch(ge, st);
        auto t1 = clk::now();
        (void)hipStreamSynchronize(st);
        t += us(t0, t1) / (50 / k + 1);
      }
    }
    std::printf(", \"graph_of_%d_launch_us\": %.3f", k, t / 100);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  }
  std::printf("}\n");
  (void)hipFree(d);
  return sink < 0;
}
