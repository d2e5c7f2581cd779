// match.hip -- appearance matcher: exact radius-bounded nearest neighbour in
// the 10-D appearance space by tiled brute force, replacing the kd-tree of
// compute_correspondences_images (vo_complete.cpp:12-49, eigen_kdtree.h:90-115,
// brute_force_search.h:22-41).
//
// Semantics kept from the reference: the larger set is searched ("tree"), the
// smaller set queries in ascending index, a hit needs d2 < radius*radius
// (strict, product evaluated in float), the closest hit wins.  Exact ties go
// to the lowest tree index (the reference's tie order is an artefact of its
// PCA partition).
//
// Exactness: d2 is accumulated exactly like the scalar reference loop --
// ((t0-q0)^2 + (t1-q1)^2) + ... one term after the other, no FMA -- so the
// running sum after 3 terms is bit-identical to the reference's intermediate
// value and, all terms being non-negative, a lower bound of the final sum.
// A candidate whose 3-term prefix already fails "prefix < best" can never pass
// "d2 < best": the early exit changes no decision.  On appearance data the
// prefix test rejects all but ~5e-4 of the pairs, so the kernel issues ~10
// instead of ~31 VALU instructions per pair.
//
// Layout: queries live in registers (QPT per thread); tree points are staged
// through LDS in tiles, 12 floats (48 B) per point so that one ds_read_b128
// broadcast fetches the 3-term prefix and two more reads the rest.  The grid
// is (query blocks) x (tree chunks); chunks merge through one 64-bit
// atomicMin per hit on key = (bits(d2) << 32) | tree index, which is
// order-independent, hence deterministic.
#include "vo_internal.h"

namespace vo {

hipError_t launch_match_compact(hipStream_t st, const unsigned long long* d_best, int nq, int tree_is_1,
                                int32_t* d_out, int* d_n_out, int* d_scratch);

constexpr int MB = 256;       // threads per workgroup
constexpr int QPT = 2;        // queries per thread
constexpr int TILE = 512;     // tree points per LDS tile (24 KiB)
constexpr int TP = 12;        // padded floats per tree point in LDS

__global__ __launch_bounds__(256) void match_init_kernel(unsigned long long* best, int nq, float r2) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) best[q] = ((unsigned long long)__float_as_uint(r2) << 32) | 0xffffffffull;
}

__global__ __launch_bounds__(MB) void match_kernel(const float* __restrict__ tree, int nt,
                                                   const float* __restrict__ qry, int nq,
                                                   int chunk, float r2,
                                                   unsigned long long* __restrict__ best) {
  __shared__ __attribute__((aligned(16))) float s_t[TILE * TP];
  const int tid = threadIdx.x;
  const int q0 = (blockIdx.x * MB + tid) * QPT;
  float q[QPT][10];
  float bd[QPT];
  int bi[QPT];
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const int qi = q0 + j < nq ? q0 + j : (nq > 0 ? nq - 1 : 0);   // clamp: result discarded
    const float2* src = reinterpret_cast<const float2*>(qry + 10 * (size_t)qi);
#pragma unroll
    for (int k = 0; k < 5; ++k) { const float2 v = src[k]; q[j][2 * k] = v.x; q[j][2 * k + 1] = v.y; }
    bd[j] = r2;           // brute_force_search.h:31
    bi[j] = -1;
  }
  const int t_begin = blockIdx.y * chunk;
  const int t_end = t_begin + chunk < nt ? t_begin + chunk : nt;
  for (int tb = t_begin; tb < t_end; tb += TILE) {
    const int cnt = t_end - tb < TILE ? t_end - tb : TILE;
    __syncthreads();
    // stage: cnt*10 contiguous floats -> 12-float records
    const float* src = tree + 10 * (size_t)tb;
    for (int f = tid; f < cnt * 10; f += MB) {
      const int p = f / 10, c = f - p * 10;
      s_t[p * TP + c] = src[f];
    }
    __syncthreads();
#pragma unroll 4
    for (int p = 0; p < cnt; ++p) {
      const float4 ta = *reinterpret_cast<const float4*>(&s_t[p * TP]);
      float pre[QPT];
      bool any = false;
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        const float d0 = ta.x - q[j][0], d1 = ta.y - q[j][1], d2 = ta.z - q[j][2];
        float s = d0 * d0;
        s += d1 * d1;
        s += d2 * d2;
        pre[j] = s;
        any = any || (s < bd[j]);
      }
      if (__builtin_expect(any, 0)) {
        const float4 tb4 = *reinterpret_cast<const float4*>(&s_t[p * TP + 4]);
        const float2 tc = *reinterpret_cast<const float2*>(&s_t[p * TP + 8]);
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
          float s = pre[j];
          float d;
          d = ta.w - q[j][3]; s += d * d;
          d = tb4.x - q[j][4]; s += d * d;
          d = tb4.y - q[j][5]; s += d * d;
          d = tb4.z - q[j][6]; s += d * d;
          d = tb4.w - q[j][7]; s += d * d;
          d = tc.x - q[j][8]; s += d * d;
          d = tc.y - q[j][9]; s += d * d;
          if (s < bd[j]) { bd[j] = s; bi[j] = tb + p; }   // brute_force_search.h:35-38
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    if (bi[j] >= 0 && q0 + j < nq) {
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(bd[j]) << 32) | (unsigned long long)(unsigned)bi[j];
      atomicMin(&best[q0 + j], key);
    }
  }
}

hipError_t launch_match(hipStream_t st, const float* d_a1, int n1, const float* d_a2, int n2,
                        float radius, int32_t* d_out_pairs, int* d_n_out,
                        unsigned long long* d_best, int* d_scratch, int n_cu) {
  const int tree_is_1 = n1 >= n2;                 // vo_complete.cpp:15-20 (ties: a1 is the tree)
  const float* tree = tree_is_1 ? d_a1 : d_a2;
  const float* qry = tree_is_1 ? d_a2 : d_a1;
  const int nt = tree_is_1 ? n1 : n2, nq = tree_is_1 ? n2 : n1;
  const float r2 = radius * radius;
  if (nq > 0) {
    hipLaunchKernelGGL(match_init_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, d_best, nq, r2);
    if (nt > 0) {
      const int qblocks = (nq + MB * QPT - 1) / (MB * QPT);
      // enough tree chunks to put ~8 workgroups on every CU, whole tiles each
      int want = (8 * (n_cu > 0 ? n_cu : 256) + qblocks - 1) / qblocks;
      const int tiles = (nt + TILE - 1) / TILE;
      if (want > tiles) want = tiles;
      if (want < 1) want = 1;
      const int tiles_per_chunk = (tiles + want - 1) / want;
      const int chunk = tiles_per_chunk * TILE;
      const int nchunks = (nt + chunk - 1) / chunk;
      hipLaunchKernelGGL(match_kernel, dim3(qblocks, nchunks), dim3(MB), 0, st, tree, nt, qry, nq,
                         chunk, r2, d_best);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_match_compact(st, d_best, nq, tree_is_1, d_out_pairs, d_n_out, d_scratch);
}

}  // namespace vo
