// Shader clock seen by a kernel that occupies ONE compute unit against one that fills the chip, and the cost of a dependent
// v_add_f32 chain in both (diagnostic).   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void chain(float* out, unsigned long long* t, int iters, float x) {
  float r = threadIdx.x;
  unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                 "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n" : "+v"(r) : "v"(x));
  }
  unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

int main() {
  float* d; unsigned long long* t; unsigned long long h[2];
  hipMalloc(&d, 1024 * 256 * 4); hipMalloc(&t, 16);
  int wall_khz = 0;
  hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  const int iters = 200000;
  for (int blocks : {1, 1, 256, 1024, 1}) {
    hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, t, iters, 1.0f);
    hipDeviceSynchronize();
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double ns = (double)h[1] / (wall_khz * 1e3) * 1e9;
    std::printf("%4d workgroup(s) of one wave: %.1f ns per dependent add, %.2f cycle-counter ticks per add, counter at %.0f MHz\n", blocks,
                ns / (8.0 * iters), (double)h[0] / (8.0 * iters), (double)h[0] / ns * 1e3);
  }
  return 0;
}
