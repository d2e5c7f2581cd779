// VALU issue rate of v_fma_f32 against v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950 (diagnostic).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_rate tools/micro/pk_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int CH = 8, ITER = 4096;

template <int MODE>
__global__ __launch_bounds__(768) void rate_kernel(float* out, float s) {
  float r = 0.f;
  if (MODE == 0) {
    float a[CH];
    for (int k = 0; k < CH; ++k) a[k] = threadIdx.x * 1e-3f + k;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
      for (int k = 0; k < CH; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(s));
    }
    for (int k = 0; k < CH; ++k) r += a[k];
  } else {
    v2f a[CH];
    v2f sv = {s, s};
    for (int k = 0; k < CH; ++k) a[k] = v2f{threadIdx.x * 1e-3f + k, 1.f};
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[k]) : "v"(sv));
        if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(sv));
        if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(sv));
      }
    }
    for (int k = 0; k < CH; ++k) r += a[k].x + a[k].y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
static void run(const char* name, int threads, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)ITER * CH * (threads / 64);           // wave-instructions per CU
  std::printf("%-14s %4d threads/CU: %.3f ms, %.2f ns per wave-instruction per SIMD\n", name, threads, ms, ms * 1e6 / (instr / 4));
}

int main() {
  float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
  for (int threads : {256, 768, 1024}) {
    run<0>("v_fma_f32", threads, d); run<1>("v_pk_fma_f32", threads, d); run<2>("v_pk_mul_f32", threads, d); run<3>("v_pk_add_f32", threads, d);
  }
  hipFree(d);
  return 0;
}
