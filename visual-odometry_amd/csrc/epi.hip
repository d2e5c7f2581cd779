// epi.hip -- the front half of the epipolar initialisation on the device (epipolar_utils.cpp:48-65,103-127,187-211):
//   epi_max_kernel    the per-axis maxima normalize() scales an image's points by (max over ALL points of the image,
//                     starting from 0: epipolar_utils.cpp:50-56);
//   epi_ata_kernel    the 45 distinct entries of A^T A, A the N x 9 system of the 8-point method (one row per
//                     correspondence: the outer product of the two normalised homogeneous points, :114-126), accumulated
//                     in double -- per workgroup, then summed over the workgroups in a fixed order (epi_sum_kernel);
//   epi_vote_kernel   the cheirality vote: how many correspondences triangulate under each of the four (R, +-t)
//                     candidates (:187-211) -- the same per-pair arithmetic as the triangulation kernel (tri_constants,
//                     triangulate_point: vo_math.h), counted, nothing written.
// What stays on the host, in double: the 9 x 9 eigen-solve, the rank-2 projection, E = K^T F K and its decomposition
// (include/vo/epipolar.hpp) -- a few microseconds of serial arithmetic.
#include "vo_internal.h"

namespace vo {

constexpr int EB = 256;

__device__ __forceinline__ int epi_rows(const int* d_n, int n_max) {
  int n = n_max;
  if (d_n) { const int m = *d_n; n = m < n ? (m < 0 ? 0 : m) : n; }
  return n;
}

// out[0..1] = max x, max y of p1, out[2..3] of p2, as float bits (zeroed by the host: "max = 0.f" of the reference;
// a positive float orders like its bit pattern, values <= 0 and NaN never replace the running maximum -- as `v > max`)
__global__ __launch_bounds__(EB) void epi_max_kernel(const float* __restrict__ p1, int n1, const float* __restrict__ p2, int n2,
                                                     unsigned* out) {
  __shared__ float s_m[4][EB / 64];
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = blockIdx.x * EB + threadIdx.x; i < n1; i += gridDim.x * EB) {
    const float2 v = reinterpret_cast<const float2*>(p1)[i];
    if (v.x > m[0]) m[0] = v.x;
    if (v.y > m[1]) m[1] = v.y;
  }
  for (int i = blockIdx.x * EB + threadIdx.x; i < n2; i += gridDim.x * EB) {
    const float2 v = reinterpret_cast<const float2*>(p2)[i];
    if (v.x > m[2]) m[2] = v.x;
    if (v.y > m[3]) m[3] = v.y;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const float o = __shfl_xor(m[k], d); if (o > m[k]) m[k] = o; }
    if ((threadIdx.x & 63) == 0) s_m[k][threadIdx.x >> 6] = m[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.f;
    for (int w = 0; w < EB / 64; ++w) if (s_m[threadIdx.x][w] > v) v = s_m[threadIdx.x][w];
    if (v > 0.f) atomicMax(&out[threadIdx.x], __float_as_uint(v));
  }
}

struct EpiArgs {
  const int32_t* pairs; int n_max; const int* d_n;
  const float* p1; int n1;
  const float* p2; int n2;
  const unsigned* maxima;     // epi_max_kernel's output
  double* partials;           // [grid][45]
  double* ata;                // [45] upper triangle, row-major: (i, j >= i)
  int* info;                  // [0] correspondences used  [1] pairs with an index outside the point arrays
  int grid;
};

__global__ __launch_bounds__(EB) void epi_ata_kernel(EpiArgs a) {
  __shared__ double s_red[EB / 64][45];
  const int n = epi_rows(a.d_n, a.n_max);
  // normalize(): x / (max_x / 2.f) - 1.f in float (epipolar_utils.cpp:58-59)
  const float hx1 = __uint_as_float(a.maxima[0]) / 2.f, hy1 = __uint_as_float(a.maxima[1]) / 2.f;
  const float hx2 = __uint_as_float(a.maxima[2]) / 2.f, hy2 = __uint_as_float(a.maxima[3]) / 2.f;
  double acc[45];
#pragma unroll
  for (int k = 0; k < 45; ++k) acc[k] = 0.0;
  int bad = 0;
  for (int i = blockIdx.x * EB + threadIdx.x; i < n; i += gridDim.x * EB) {
    const int2 pr = reinterpret_cast<const int2*>(a.pairs)[i];
    if (pr.x < 0 || pr.x >= a.n1 || pr.y < 0 || pr.y >= a.n2) { ++bad; continue; }
    const float2 q1 = reinterpret_cast<const float2*>(a.p1)[pr.x];
    const float2 q2 = reinterpret_cast<const float2*>(a.p2)[pr.y];
    const double d1[3] = {(double)(q1.x / hx1 - 1.f), (double)(q1.y / hy1 - 1.f), 1.0};
    const double d2[3] = {(double)(q2.x / hx2 - 1.f), (double)(q2.y / hy2 - 1.f), 1.0};
    double row[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) row[3 * r + c] = d1[r] * d2[c];                       // :124-125
    int k = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int c = r; c < 9; ++c) acc[k++] += row[r] * row[c];
  }
  if (bad) atomicAdd(&a.info[1], bad);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 45; ++k) {
    double v = acc[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) s_red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 45) {
    double v = 0.0;
    for (int w = 0; w < EB / 64; ++w) v += s_red[w][threadIdx.x];
    a.partials[(size_t)blockIdx.x * 45 + threadIdx.x] = v;
  }
}

__global__ void epi_sum_kernel(EpiArgs a) {
  const int k = threadIdx.x;
  if (k < 45) {
    double v = 0.0;
    for (int b = 0; b < a.grid; ++b) v += a.partials[(size_t)b * 45 + k];
    a.ata[k] = v;
  }
  if (k == 63) a.info[0] = epi_rows(a.d_n, a.n_max);
}

struct VoteArgs {
  float K[9];
  Pose X[4];
  const int32_t* pairs; int n_max; const int* d_n;
  const float* p1; int n1;
  const float* p2; int n2;
  int* counts;                // [4], zeroed by the host
};

__global__ __launch_bounds__(EB) void epi_vote_kernel(VoteArgs a) {
  __shared__ TriConst s_c[4];
  __shared__ int s_cnt[4];
  if (threadIdx.x < 4) { s_c[threadIdx.x] = tri_constants(a.K, a.X[threadIdx.x]); s_cnt[threadIdx.x] = 0; }      // utils.cpp:79-82
  __syncthreads();
  const int n = epi_rows(a.d_n, a.n_max);
  int cnt[4] = {0, 0, 0, 0};
  for (int i = blockIdx.x * EB + threadIdx.x; i < n; i += gridDim.x * EB) {
    const int2 pr = reinterpret_cast<const int2*>(a.pairs)[i];
    if (pr.x < 0 || pr.x >= a.n1 || pr.y < 0 || pr.y >= a.n2) continue;
    const float2 q1 = reinterpret_cast<const float2*>(a.p1)[pr.x];
    const float2 q2 = reinterpret_cast<const float2*>(a.p2)[pr.y];
    const float h1[3] = {q1.x, q1.y, 1.f}, h2[3] = {q2.x, q2.y, 1.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d1[3], d2[3], p[3];
      mat3_vec(s_c[k].iK, 3, h1, d1);          // utils.cpp:60-62 (v1), as tri_eval
      mat3_vec(s_c[k].iRiK, 3, h2, d2);
      cnt[k] += triangulate_point(d1, d2, s_c[k].t, p) ? 1 : 0;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int v = cnt[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_cnt[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&a.counts[threadIdx.x], s_cnt[threadIdx.x]);
}

size_t epi_workspace_bytes() { return 16 + 16 + 32 + sizeof(double) * 45 + sizeof(double) * 45 * 512; }

// ws (zeroed here): [0,16) maxima  [16,32) info  [32,64) vote counts  [64, 64+360) ata  then the partials
hipError_t launch_epi_front(hipStream_t st, const int32_t* d_pairs, int n_max, const int* d_n, const float* d_p1, int n1,
                            const float* d_p2, int n2, void* ws) {
  hipError_t e = hipMemsetAsync(ws, 0, 64, st);
  if (e != hipSuccess) return e;
  char* w = static_cast<char*>(ws);
  unsigned* maxima = reinterpret_cast<unsigned*>(w);
  int big = n1 > n2 ? n1 : n2;
  int g = (big + EB - 1) / EB; if (g < 1) g = 1; if (g > 512) g = 512;
  hipLaunchKernelGGL(epi_max_kernel, dim3(g), dim3(EB), 0, st, d_p1, n1, d_p2, n2, maxima);
  EpiArgs a;
  a.pairs = d_pairs; a.n_max = n_max; a.d_n = d_n; a.p1 = d_p1; a.n1 = n1; a.p2 = d_p2; a.n2 = n2;
  a.maxima = maxima; a.info = reinterpret_cast<int*>(w + 16);
  a.ata = reinterpret_cast<double*>(w + 64); a.partials = a.ata + 45;
  a.grid = (n_max + EB - 1) / EB; if (a.grid < 1) a.grid = 1; if (a.grid > 512) a.grid = 512;
  hipLaunchKernelGGL(epi_ata_kernel, dim3(a.grid), dim3(EB), 0, st, a);
  hipLaunchKernelGGL(epi_sum_kernel, dim3(1), dim3(64), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_epi_vote(hipStream_t st, const float K[9], const Pose X[4], const int32_t* d_pairs, int n_max, const int* d_n,
                           const float* d_p1, int n1, const float* d_p2, int n2, void* ws) {
  VoteArgs a;
  for (int k = 0; k < 9; ++k) a.K[k] = K[k];
  for (int k = 0; k < 4; ++k) a.X[k] = X[k];
  a.pairs = d_pairs; a.n_max = n_max; a.d_n = d_n; a.p1 = d_p1; a.n1 = n1; a.p2 = d_p2; a.n2 = n2;
  a.counts = reinterpret_cast<int*>(static_cast<char*>(ws) + 32);
  int g = (n_max + EB - 1) / EB; if (g < 1) g = 1; if (g > 1024) g = 1024;
  hipLaunchKernelGGL(epi_vote_kernel, dim3(g), dim3(EB), 0, st, a);
  return hipGetLastError();
}

}  // namespace vo
