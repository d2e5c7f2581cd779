// valu_rate.hip -- microbenchmark: sustained issue rate of v_fma_f32 vs
// v_pk_fma_f32 on gfx950 as a function of waves per SIMD.  Diagnostic tool,
// not part of the library.   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
  float2v p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, p4 = {r1, r0}, p5 = {r3, r2}, p6 = {r5, r4}, p7 = {r7, r6};
  const float2v a2 = {a, a}, b2 = {b, b};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
      } else if (MODE == 1) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(a2), "v"(b2));
      } else if (MODE == 2) {   // mul + add, dependent pairs (like contract-off code)
        asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                     "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
      } else {                  // fully dependent chain
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(r0) : "v"(a), "v"(b));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char* names[4] = {"v_fma_f32 x8 indep", "v_pk_fma_f32 x8 indep", "v_mul/v_add indep", "v_fma_f32 dependent"};
  for (int mode = 0; mode < 4; ++mode)
    for (int threads : {256, 512, 1024}) {      // 1, 2, 4 waves per SIMD, one workgroup per CU
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double wave_instr_per_simd = (double)iters * 64 * (threads / 256);   // 64 instrs per iter per wave
      const double ns_per_instr = ms * 1e6 / wave_instr_per_simd;
      const double lanes = mode == 1 ? 128.0 : 64.0, flop = (mode == 2) ? 1.0 : 2.0;
      printf("%-24s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)  chip %.1f TFLOP/s\n",
             names[mode], threads / 256, ms, ns_per_instr, ns_per_instr * 2.4,
             256.0 * 4 * lanes * flop / ns_per_instr / 1e3);
    }
  return 0;
}
