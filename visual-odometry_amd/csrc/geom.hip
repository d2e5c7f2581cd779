// geom.hip -- single-pass, HBM-bound kernels around the solver:
//   Camera::projectPoints           (camera.cpp:16-37)
//   Isometry3f * point set          (PointCloud.h:77-82)
//   triangulate_points v1/v2/v3     (utils.cpp:51-134)
//   extract_correspondences_world   (vo_complete.cpp:52-66)
//
// Variable-length outputs keep the reference's order (stable compaction):
//   count kernel   : every 256-item workgroup counts its survivors
//   scan kernel    : one workgroup turns the counts into exclusive offsets and
//                    the total (left in device memory: no host round trip)
//   scatter kernel : re-evaluates the (cheap, deterministic) predicate, ranks
//                    survivors inside the workgroup by ballot/popcount and
//                    writes them at offset + rank.
// Nothing is staged in a temporary array: re-evaluating ~60 flops is cheaper
// than writing and re-reading 12-44 B per item.
#include <limits.h>

#include <stdlib.h>

#include "vo_internal.h"
#include "chain_scan.h"

namespace vo {

constexpr int CB = 256;  // items per compaction workgroup

// per frame: the chain state of the single-pass compactions (chain_scan.h: 8-byte words) -- which also covers the
// per-workgroup counts of the count / scan / scatter form (4-byte) -- plus 64 bytes
size_t compaction_scratch_ints(int n) { return 2 * chain_words((n + CB - 1) / CB) + 16; }
static inline size_t chain_stride(int n) { return compaction_scratch_ints(n) / 2; }     // 8-byte words per frame

// The compactions run as count / scan / scatter (two visits of every item, a scan launch in between).  VO_ONE_PASS=1 in the
// environment (read at every launch: a test flips it inside one process) selects the single-pass form instead -- a chained
// scan with decoupled look-back, chain_scan.h -- which round 5 built to spare the second visit and MEASURED SLOWER on this
// path: per 200 x 50k frames the triangulation takes 335-394 us against 297 (count 111 + scatter 186), the join with the
// solver's gather 247-295 against 222, the whole call 2.57-2.97 ms against 2.54-2.56 (1, 2, 4, 8, 12, 16 rows of 256 items per
// workgroup: 3.57 / 2.97 / 2.68 / 2.57 / 2.61 / 2.64 ms; same box, tools/ab_frames.sh).  Why: the two-pass kernels already
// move their ACTUAL bytes (algorithmic + the 24 B per item they hand from pass to pass) at 5.2 TB/s, so a perfect single
// pass could save 19 % of their time at most; and a chained workgroup holds its slot through three memory-side round trips
// (ticket, published count, look-back: agent-scope atomics, the only hand-off that is correct across XCDs) with nothing in
// flight, which costs more memory-level parallelism than the second visit costs bytes.  DESIGN.md section 8.
static bool two_pass() {
  const char* e = getenv("VO_ONE_PASS");
  return !(e && e[0] == '1');
}

// exclusive rank of `flag` inside a 256-thread workgroup; total = survivors
__device__ __forceinline__ int block_rank(bool flag, int* s_wave, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) s_wave[wave] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < CB / 64; ++w) {
    const int c = s_wave[w];
    if (w < wave) off += c;
    tot += c;
  }
  total = tot;
  return off + before;
}

// Small frames -- every frame of the reference's dataset holds 14..127 points -- are compacted by ONE workgroup per frame: it walks
// the items 256 at a time with the running count in a register, so count, scan and scatter are one launch instead of three
// (the join: table fill and build too -- one instead of five; the table lives in LDS).  Same ranks, same order, same output.
#ifndef VO_SMALL_N
#define VO_SMALL_N 2048
#endif
constexpr int SMALL_N = VO_SMALL_N;   // (-DVO_SMALL_N=-1: the general kernels for every size)

__device__ __forceinline__ int clamp_count(const int* d_n, int n_max) {
  int n = n_max;
  if (d_n) { const int m = *d_n; n = m < n_max ? (m < 0 ? 0 : m) : n_max; }
  return n;
}

// One workgroup per frame (blockIdx.x): offsets[b] = sum(counts[0..b-1]) in place, *total = sum.
__global__ __launch_bounds__(1024) void scan_counts_kernel(int* counts, int nb, int* total,
                                                           int* total2, size_t counts_stride) {
  counts += blockIdx.x * counts_stride;
  total += blockIdx.x;
  if (total2) total2 += blockIdx.x;
  __shared__ int s_w[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + tid;
    const int v = i < nb ? counts[i] : 0;
    int incl = v;   // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int c = s_w[w];
      if (w < wave) woff += c;
      tot += c;
    }
    const int carry = s_carry;
    if (i < nb) counts[i] = carry + woff + incl - v;
    __syncthreads();
    if (tid == 0) s_carry = carry + tot;
    __syncthreads();
  }
  if (tid == 0) {
    *total = s_carry;
    if (total2) *total2 = s_carry;
  }
}

hipError_t launch_scan(hipStream_t st, int* counts, int nb, int* total, int* total2, int n_frames, size_t counts_stride) {
  hipLaunchKernelGGL(scan_counts_kernel, dim3(n_frames), dim3(1024), 0, st, counts, nb, total, total2, counts_stride);
  return hipGetLastError();
}

// ---- projectPoints ------------------------------------------------------------
struct ProjArgs {
  CamK cam;
  Pose T;
  const float* world;
  int n;
  int keep_indices;
  float* out_uv;
  int* counts;   // per workgroup
};

__device__ __forceinline__ bool proj_eval(const ProjArgs& a, int i, float& u, float& v) {
  float pc[3], ph[3], inv;
  const float* p = a.world + 3 * (size_t)i;
  return project_point(a.cam, a.T, p[0], p[1], p[2], u, v, pc, ph, inv);
}

// keep_indices: writes (u,v) or (-1,-1) in place and counts; else counts only
__global__ __launch_bounds__(CB) void project_count_kernel(ProjArgs a) {
  __shared__ int s_wave[CB / 64];
  const int i = blockIdx.x * CB + threadIdx.x;
  bool ok = false;
  float u = -1.f, v = -1.f;
  if (i < a.n) {
    ok = proj_eval(a, i, u, v);
    if (a.keep_indices) {
      float2 o = ok ? make_float2(u, v) : make_float2(-1.f, -1.f);   // camera.cpp:30
      *reinterpret_cast<float2*>(a.out_uv + 2 * (size_t)i) = o;
    }
  }
  int total;
  block_rank(ok, s_wave, total);
  if (threadIdx.x == 0) a.counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(CB) void project_scatter_kernel(ProjArgs a) {
  __shared__ int s_wave[CB / 64];
  const int i = blockIdx.x * CB + threadIdx.x;
  bool ok = false;
  float u = 0.f, v = 0.f;
  if (i < a.n) ok = proj_eval(a, i, u, v);
  int total;
  const int r = block_rank(ok, s_wave, total);
  if (ok) {
    const size_t dst = (size_t)a.counts[blockIdx.x] + r;
    *reinterpret_cast<float2*>(a.out_uv + 2 * dst) = make_float2(u, v);
  }
}

__global__ void project_finish_kernel(int* d_counts, int n, int keep_indices) {
  // d_counts[1] = n_inside (written by the scan); d_counts[0] = n_out
  d_counts[0] = keep_indices ? n : d_counts[1];
}

hipError_t launch_project_points(hipStream_t st, const CamK& cam, const Pose& T, const float* d_world,
                                 int n, int keep_indices, float* d_out_uv, int* d_counts,
                                 int* d_scratch) {
  const int nb = (n + CB - 1) / CB;
  ProjArgs a{cam, T, d_world, n, keep_indices, d_out_uv, d_scratch};
  if (nb > 0) hipLaunchKernelGGL(project_count_kernel, dim3(nb), dim3(CB), 0, st, a);
  hipError_t e = launch_scan(st, d_scratch, nb, d_counts + 1);
  if (e != hipSuccess) return e;
  if (!keep_indices && nb > 0) hipLaunchKernelGGL(project_scatter_kernel, dim3(nb), dim3(CB), 0, st, a);
  hipLaunchKernelGGL(project_finish_kernel, dim3(1), dim3(1), 0, st, d_counts, n, keep_indices);
  return hipGetLastError();
}

// ---- rigid transform ------------------------------------------------------------
__global__ __launch_bounds__(256) void transform_kernel(Pose T, const float* __restrict__ in, int n_max,
                                                        const int* __restrict__ d_n,
                                                        float* __restrict__ out) {
  const int n = clamp_count(d_n, n_max);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = in + 3 * (size_t)i;
    float ox, oy, oz;
    pose_apply(T, p[0], p[1], p[2], ox, oy, oz);
    float* o = out + 3 * (size_t)i;
    o[0] = ox; o[1] = oy; o[2] = oz;
  }
}

// same with the isometry read from device memory (the pose a previous solve left there)
__global__ __launch_bounds__(256) void transform_devpose_kernel(const float* __restrict__ T16, const float* __restrict__ in,
                                                                int n_max, const int* __restrict__ d_n,
                                                                float* __restrict__ out) {
  float t[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) t[k] = T16[k];
  const Pose T = pose_from_T16(t);
  const int n = clamp_count(d_n, n_max);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = in + 3 * (size_t)i;
    float ox, oy, oz;
    pose_apply(T, p[0], p[1], p[2], ox, oy, oz);
    float* o = out + 3 * (size_t)i;
    o[0] = ox; o[1] = oy; o[2] = oz;
  }
}

// batched: frame f = blockIdx.y reads T16[f] (column-major 4x4; null: identity) and in/out + f*stride points
__global__ __launch_bounds__(256) void transform_batch_kernel(const float* __restrict__ T16, const float* __restrict__ in,
                                                              int n, size_t stride, float* __restrict__ out) {
  const int f = blockIdx.y;
  Pose T;
  if (T16) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = T16[16 * (size_t)f + k];
    T = pose_from_T16(t);
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) T.R[k] = (k % 4 == 0) ? 1.f : 0.f;
    T.t[0] = T.t[1] = T.t[2] = 0.f;
  }
  in += 3 * f * stride;
  out += 3 * f * stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* p = in + 3 * (size_t)i;
    float ox, oy, oz;
    pose_apply(T, p[0], p[1], p[2], ox, oy, oz);
    float* o = out + 3 * (size_t)i;
    o[0] = ox; o[1] = oy; o[2] = oz;
  }
}

hipError_t launch_transform_batch(hipStream_t st, const float* d_T16, const float* d_in, int n, size_t stride,
                                  float* d_out, int n_frames) {
  if (n <= 0 || n_frames <= 0) return hipSuccess;
  int grid = (n + 255) / 256;
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(transform_batch_kernel, dim3(grid, n_frames), dim3(256), 0, st, d_T16, d_in, n, stride, d_out);
  return hipGetLastError();
}

hipError_t launch_transform_points(hipStream_t st, const Pose& T, const float* d_in, int n,
                                   const int* d_n, float* d_out) {
  if (n <= 0) return hipSuccess;
  int grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(transform_kernel, dim3(grid), dim3(256), 0, st, T, d_in, n, d_n, d_out);
  return hipGetLastError();
}

hipError_t launch_transform_points_devpose(hipStream_t st, const float* d_T16, const float* d_in, int n,
                                           const int* d_n, float* d_out) {
  if (n <= 0) return hipSuccess;
  int grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(transform_devpose_kernel, dim3(grid), dim3(256), 0, st, d_T16, d_in, n, d_n, d_out);
  return hipGetLastError();
}

// ---- triangulation ----------------------------------------------------------------
struct TriArgs {
  float K[9];
  Pose X;               // used when d_X16 == nullptr
  const float* d_X16;   // pose in device memory (e.g. the solver's result)
  const int32_t* pairs;
  int n_max;
  const int* d_n;
  const float* p1; int n1;
  const float* p2; int n2;
  const float* app2;
  float* out_xyz;
  int32_t* out_pairs;
  float* out_app;
  int* counts;
  // what the counting pass keeps for the writing pass: every pair's point and one bit per pair ("survives")
  float* tmp_xyz;                   // [n_frames][n_max][3]
  unsigned long long* tmp_ok;       // [n_frames][(n_max + 63) / 64]
  // batched use: strides in pairs / points per frame (0 for a single frame); nb workgroups for each of n_frames frames
  size_t pairs_stride, p1_stride, p2_stride, out_stride, counts_stride;
  int nb, n_frames;
  unsigned long long* chain;        // single-pass form: [n_frames][chain_stride] status words (zeroed before the launch)
  size_t chain_stride;
};

// shifts every per-frame pointer of `a` to frame f
__device__ __forceinline__ TriArgs tri_frame(TriArgs a, int frame) {
  const size_t f = (size_t)frame;
  if (a.d_X16) a.d_X16 += 16 * f;
  a.pairs += 2 * f * a.pairs_stride;
  if (a.d_n) a.d_n += f;
  a.p1 += 2 * f * a.p1_stride;
  a.p2 += 2 * f * a.p2_stride;
  if (a.app2) a.app2 += 10 * f * a.p2_stride;
  a.out_xyz += 3 * f * a.out_stride;
  if (a.out_pairs) a.out_pairs += 2 * f * a.out_stride;
  if (a.out_app) a.out_app += 10 * f * a.out_stride;
  a.counts += f * a.counts_stride;
  if (a.tmp_xyz) { a.tmp_xyz += 3 * f * (size_t)a.n_max; a.tmp_ok += f * (size_t)((a.n_max + 63) / 64); }
  a.chain += f * a.chain_stride;
  return a;
}

__device__ __forceinline__ void tri_setup(const TriArgs& a, TriConst* s_c) {
  if (threadIdx.x == 0) {
    Pose X = a.X;
    if (a.d_X16) {
      float T16[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) T16[k] = a.d_X16[k];
      X = pose_from_T16(T16);
    }
    *s_c = tri_constants(a.K, X);      // utils.cpp:79-82
  }
  __syncthreads();
}

__device__ __forceinline__ bool tri_eval(const TriArgs& a, const TriConst& c, int k, int& i2,
                                         float p[3]) {
  const int i1 = a.pairs[2 * (size_t)k];
  i2 = a.pairs[2 * (size_t)k + 1];
  if (i1 < 0 || i1 >= a.n1 || i2 < 0 || i2 >= a.n2) return false;   // out of range: dropped
  const float2 q1 = *reinterpret_cast<const float2*>(a.p1 + 2 * (size_t)i1);
  const float2 q2 = *reinterpret_cast<const float2*>(a.p2 + 2 * (size_t)i2);
  const float h1[3] = {q1.x, q1.y, 1.f}, h2[3] = {q2.x, q2.y, 1.f};
  float d1[3], d2[3];
  mat3_vec(c.iK, 3, h1, d1);     // utils.cpp:89-91
  mat3_vec(c.iRiK, 3, h2, d2);   // utils.cpp:92-94
  return triangulate_point(d1, d2, c.t, p);
}

__global__ __launch_bounds__(CB) void tri_count_kernel(TriArgs a0) {
  const FrameBlock fb = frame_block(a0.nb, a0.n_frames);
  if (!fb.live) return;
  const TriArgs a = tri_frame(a0, fb.f);
  __shared__ TriConst s_c;
  __shared__ int s_wave[CB / 64];
  tri_setup(a, &s_c);
  const int n = clamp_count(a.d_n, a.n_max);
  const int k = fb.b * CB + threadIdx.x;
  bool ok = false;
  float p[3] = {0.f, 0.f, 0.f};
  if (k < n) { int i2; ok = tri_eval(a, s_c, k, i2, p); }
  // kept for the writing pass (coalesced), which then neither gathers the image points nor triangulates again
  if (k < a.n_max) { a.tmp_xyz[3 * (size_t)k] = p[0]; a.tmp_xyz[3 * (size_t)k + 1] = p[1]; a.tmp_xyz[3 * (size_t)k + 2] = p[2]; }
  const unsigned long long m = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && k < a.n_max) a.tmp_ok[k >> 6] = m;
  int total;
  block_rank(ok, s_wave, total);
  if (threadIdx.x == 0) a.counts[fb.b] = total;
}

__global__ __launch_bounds__(CB) void tri_scatter_kernel(TriArgs a0) {
  const FrameBlock fb = frame_block(a0.nb, a0.n_frames);
  if (!fb.live) return;
  const TriArgs a = tri_frame(a0, fb.f);
  __shared__ int s_wave[CB / 64];
  const int k = fb.b * CB + threadIdx.x;
  // the counting pass left every pair's point and verdict (tri_count_kernel): nothing is gathered or triangulated twice
  bool ok = false;
  int i2 = 0;
  float p[3] = {0.f, 0.f, 0.f};
  if (k < a.n_max) {
    ok = (a.tmp_ok[k >> 6] >> (k & 63)) & 1ull;
    if (ok) {
      i2 = a.pairs[2 * (size_t)k + 1];
      p[0] = a.tmp_xyz[3 * (size_t)k]; p[1] = a.tmp_xyz[3 * (size_t)k + 1]; p[2] = a.tmp_xyz[3 * (size_t)k + 2];
    }
  }
  int total;
  const int r = block_rank(ok, s_wave, total);
  __shared__ int s_src[CB];
  if (ok) {
    const size_t dst = (size_t)a.counts[fb.b] + r;
    a.out_xyz[3 * dst] = p[0]; a.out_xyz[3 * dst + 1] = p[1]; a.out_xyz[3 * dst + 2] = p[2];
    if (a.out_pairs) reinterpret_cast<int2*>(a.out_pairs)[dst] = make_int2(i2, (int)dst);     // utils.cpp:97
    s_src[r] = i2;
  }
  __syncthreads();
  if (a.out_app && a.app2) {                                                                 // utils.cpp:127
    // appearance copy-through, cooperatively: consecutive threads move consecutive 8-byte pieces of the survivors' rows
    // (coalesced stores, 40-byte gathers)
    const float2* app = reinterpret_cast<const float2*>(a.app2);
    float2* o = reinterpret_cast<float2*>(a.out_app) + 5 * (size_t)a.counts[fb.b];
    for (int j = threadIdx.x; j < 5 * total; j += CB) {
      const int pt = j / 5;
      o[j] = app[5 * (size_t)s_src[pt] + (j - 5 * pt)];
    }
  }
}

// ---- single-pass compaction: several rows of CB items per workgroup ----------------------------------------------------
// A workgroup of the single-pass kernels owns a CHUNK of OP_ITEMS rows of CB consecutive items (thread t takes item
// t of every row: coalesced), because what a chained scan adds to a workgroup's life -- the ticket, the published count,
// the look-back: three memory-side round trips -- is per workgroup, and these kernels are bound by the number of
// workgroups in flight, not by bytes (one row per workgroup, 196 chunks per 50k frame: 857 us per 200 frames where
// count + scatter took 297).  Eight rows: 25 chunks per frame, one look-back window, eight independent gathers per lane.
#ifndef VO_OP_ITEMS
#define VO_OP_ITEMS 8
#endif
constexpr int OP_ITEMS = VO_OP_ITEMS;
constexpr int OP_CHUNK = CB * OP_ITEMS;

// ranks of a chunk: item (row j, thread t) with flag ok[j] gets the number of flagged items in front of it in item
// order (rows before j, then threads before t).  s_cnt: OP_ITEMS * (CB / 64) ints.  One barrier.
__device__ __forceinline__ void chunk_rank(const bool ok[OP_ITEMS], int* s_cnt, int rank[OP_ITEMS], int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int before[OP_ITEMS];
#pragma unroll
  for (int j = 0; j < OP_ITEMS; ++j) {
    const unsigned long long m = __ballot(ok[j]);
    before[j] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[j * (CB / 64) + wave] = __popcll(m);
  }
  __syncthreads();
  int run = 0;
#pragma unroll
  for (int j = 0; j < OP_ITEMS; ++j) {
#pragma unroll
    for (int w = 0; w < CB / 64; ++w) {
      if (w == wave) rank[j] = run + before[j];
      run += s_cnt[j * (CB / 64) + w];
    }
  }
  total = run;
}

// The triangulation in ONE pass (chain_scan.h): triangulate, rank inside the chunk, learn the chunk's offset from its
// predecessors' published counts, write.  No point is stored for a second pass, no pair is read twice; the appearance
// rows of the survivors are requested BEFORE the look-back (their addresses do not depend on the offset) and stored after it.
__global__ __launch_bounds__(CB) void tri_onepass_kernel(TriArgs a0, int* d_n_out) {
  const FrameBlock fb = frame_block(a0.nb, a0.n_frames);       // the frame (and with it the XCD); the chunk is the ticket
  if (!fb.live) return;
  const TriArgs a = tri_frame(a0, fb.f);
  __shared__ TriConst s_c;
  __shared__ int s_cnt[OP_ITEMS * (CB / 64)];
  __shared__ int s_src[OP_CHUNK];
  __shared__ int s_b, s_excl;
  if (threadIdx.x == 0) s_b = chain_take_ticket(a.chain);
  tri_setup(a, &s_c);                                          // (its barrier publishes s_b too)
  const int b = s_b;
  const int n = clamp_count(a.d_n, a.n_max);
  bool ok[OP_ITEMS];
  int i2[OP_ITEMS], rank[OP_ITEMS];
  float p[OP_ITEMS][3];
#pragma unroll
  for (int j = 0; j < OP_ITEMS; ++j) {
    const int k = b * OP_CHUNK + j * CB + threadIdx.x;
    ok[j] = false; i2[j] = 0; p[j][0] = p[j][1] = p[j][2] = 0.f;
    if (k < n) ok[j] = tri_eval(a, s_c, k, i2[j], p[j]);
  }
  int total;
  chunk_rank(ok, s_cnt, rank, total);
#pragma unroll
  for (int j = 0; j < OP_ITEMS; ++j) if (ok[j]) s_src[rank[j]] = i2[j];
  __syncthreads();
  constexpr int PIECES = 5;                                    // 8-byte pieces of a 40-byte appearance row
  float2 piece[PIECES * OP_ITEMS];
  const bool with_app = a.out_app && a.app2;
#ifndef VO_TRI_LATE_APP
  if (with_app) {
    const float2* app = reinterpret_cast<const float2*>(a.app2);
#pragma unroll
    for (int q = 0; q < PIECES * OP_ITEMS; ++q) {
      const int j = threadIdx.x + q * CB;
      // (clamped, unconditional: a predicated load makes the compiler wait for all of them at the first use)
      const int jj = j < PIECES * total ? j : 0;
      const int pt = jj / PIECES;
      piece[q] = total > 0 ? app[PIECES * (size_t)s_src[pt] + (jj - PIECES * pt)] : make_float2(0.f, 0.f);
    }
  }
#endif
  if (threadIdx.x < 64) {
    const int e = chain_lookback(a.chain, b, total);
    if (threadIdx.x == 0) s_excl = e;
  }
  __syncthreads();
  const int excl = s_excl;
  if (excl < 0) return;                                        // the look-back gave up (chain word 1 is set): nothing is written
#pragma unroll
  for (int j = 0; j < OP_ITEMS; ++j) {
    if (ok[j]) {
      const size_t dst = (size_t)excl + rank[j];
      a.out_xyz[3 * dst] = p[j][0]; a.out_xyz[3 * dst + 1] = p[j][1]; a.out_xyz[3 * dst + 2] = p[j][2];
      if (a.out_pairs) reinterpret_cast<int2*>(a.out_pairs)[dst] = make_int2(i2[j], (int)dst);     // utils.cpp:97
    }
  }
  if (with_app) {                                                                              // utils.cpp:127
    float2* o = reinterpret_cast<float2*>(a.out_app) + PIECES * (size_t)excl;
#ifdef VO_TRI_LATE_APP
    const float2* app = reinterpret_cast<const float2*>(a.app2);
    for (int j = threadIdx.x; j < PIECES * total; j += CB) {
      const int pt = j / PIECES;
      o[j] = app[PIECES * (size_t)s_src[pt] + (j - PIECES * pt)];
    }
    (void)piece;
#else
#pragma unroll
    for (int q = 0; q < PIECES * OP_ITEMS; ++q) {
      const int j = threadIdx.x + q * CB;
      if (j < PIECES * total) o[j] = piece[q];
    }
#endif
  }
  if (b == a.nb - 1 && threadIdx.x == 0) d_n_out[fb.f] = excl + total;
}

__global__ __launch_bounds__(CB) void tri_small_kernel(TriArgs a0, int* d_n_out) {
  const int f = blockIdx.x;
  const TriArgs a = tri_frame(a0, f);
  __shared__ TriConst s_c;
  __shared__ int s_wave[CB / 64];
  __shared__ int s_src[CB];
  tri_setup(a, &s_c);
  const int n = clamp_count(a.d_n, a.n_max);
  int first = 0;
  for (int base = 0; base < n; base += CB) {
    const int k = base + threadIdx.x;
    bool ok = false;
    int i2 = 0;
    float p[3] = {0.f, 0.f, 0.f};
    if (k < n) ok = tri_eval(a, s_c, k, i2, p);
    int total;
    const int r = block_rank(ok, s_wave, total);
    if (ok) {
      const size_t dst = (size_t)first + r;
      a.out_xyz[3 * dst] = p[0]; a.out_xyz[3 * dst + 1] = p[1]; a.out_xyz[3 * dst + 2] = p[2];
      if (a.out_pairs) reinterpret_cast<int2*>(a.out_pairs)[dst] = make_int2(i2, (int)dst);     // utils.cpp:97
      s_src[r] = i2;
    }
    __syncthreads();
    if (a.out_app && a.app2) {                                                                 // utils.cpp:127
      const float2* app = reinterpret_cast<const float2*>(a.app2);
      float2* o = reinterpret_cast<float2*>(a.out_app) + 5 * (size_t)first;
      for (int j = threadIdx.x; j < 5 * total; j += CB) {
        const int pt = j / 5;
        o[j] = app[5 * (size_t)s_src[pt] + (j - 5 * pt)];
      }
    }
    first += total;
    __syncthreads();                                   // s_wave and s_src are reused by the next 256 items
  }
  if (threadIdx.x == 0) d_n_out[f] = first;
}

// bytes of d_scratch the join needs: the compaction's counts, then one looked-up index per image pair
size_t join_scratch_bytes(int n_img, int n_frames) {
  return (sizeof(int) * (compaction_scratch_ints(n_img) + (size_t)n_img) + sizeof(unsigned long long) * (size_t)((n_img + 63) / 64)) * (size_t)n_frames + 16;
}

// bytes of d_scratch the triangulation needs: the compaction's counts, then one bit and one point per pair
size_t triangulate_scratch_bytes(int n, int n_frames) {
  return (sizeof(int) * compaction_scratch_ints(n) + sizeof(unsigned long long) * (size_t)((n + 63) / 64) + 12 * (size_t)n) * (size_t)n_frames + 16;
}

// n_frames > 1: frame f uses d_X16 + 16f, pairs + f*pairs_stride, d_n[f], p1/p2/app2 + f*stride, writes
// out_* + f*out_stride and d_n_out[f]; d_scratch holds n_frames * compaction_scratch_ints(n) ints.
hipError_t launch_triangulate_batch(hipStream_t st, const float K[9], const Pose* X_host, const float* d_X16,
                                    const int32_t* d_pairs, int n, const int* d_n, const float* d_p1, int n1,
                                    const float* d_p2, int n2, const float* d_app2, float* d_out_xyz,
                                    int32_t* d_out_pairs, float* d_out_app, int* d_n_out, int* d_scratch, int n_frames,
                                    size_t pairs_stride, size_t p1_stride, size_t p2_stride, size_t out_stride) {
  TriArgs a;
  for (int k = 0; k < 9; ++k) a.K[k] = K[k];
  if (X_host) a.X = *X_host;
  else { for (int k = 0; k < 9; ++k) a.X.R[k] = (k % 4 == 0) ? 1.f : 0.f; a.X.t[0] = a.X.t[1] = a.X.t[2] = 0.f; }
  a.d_X16 = d_X16;
  a.pairs = d_pairs; a.n_max = n; a.d_n = d_n;
  a.p1 = d_p1; a.n1 = n1; a.p2 = d_p2; a.n2 = n2; a.app2 = d_app2;
  a.out_xyz = d_out_xyz; a.out_pairs = d_out_pairs; a.out_app = d_out_app;
  a.counts = d_scratch;
  a.tmp_ok = nullptr; a.tmp_xyz = nullptr;
  a.chain = reinterpret_cast<unsigned long long*>(d_scratch);
  a.chain_stride = chain_stride(n);
  if (two_pass()) {
    size_t off = sizeof(int) * compaction_scratch_ints(n) * (size_t)n_frames;   // triangulate_scratch_bytes
    off = (off + 15) & ~(size_t)15;
    char* t = reinterpret_cast<char*>(d_scratch) + off;
    a.tmp_ok = reinterpret_cast<unsigned long long*>(t);
    a.tmp_xyz = reinterpret_cast<float*>(t + sizeof(unsigned long long) * (size_t)((n + 63) / 64) * (size_t)n_frames);
  }
  const bool batched = n_frames > 1;
  a.pairs_stride = batched ? pairs_stride : 0; a.p1_stride = batched ? p1_stride : 0;
  a.p2_stride = batched ? p2_stride : 0; a.out_stride = batched ? out_stride : 0;
  a.counts_stride = batched ? compaction_scratch_ints(n) : 0;
  const int nb = (n + CB - 1) / CB;
  a.nb = nb; a.n_frames = n_frames;
  if (n <= SMALL_N) {
    hipLaunchKernelGGL(tri_small_kernel, dim3(n_frames), dim3(CB), 0, st, a, d_n_out);
    return hipGetLastError();
  }
  if (nb == 0) return hipMemsetAsync(d_n_out, 0, sizeof(int) * (size_t)n_frames, st);
  if (!two_pass()) {
    hipError_t e = hipMemsetAsync(d_scratch, 0, sizeof(int) * compaction_scratch_ints(n) * (size_t)n_frames, st);
    if (e != hipSuccess) return e;
    a.nb = (n + OP_CHUNK - 1) / OP_CHUNK;                       // chunks of OP_ITEMS rows
    hipLaunchKernelGGL(tri_onepass_kernel, frame_grid(a.nb, n_frames), dim3(CB), 0, st, a, d_n_out);
    return hipGetLastError();
  }
  if (nb > 0) hipLaunchKernelGGL(tri_count_kernel, frame_grid(nb, n_frames), dim3(CB), 0, st, a);
  hipError_t e = launch_scan(st, d_scratch, nb, d_n_out, nullptr, n_frames, a.counts_stride);
  if (e != hipSuccess) return e;
  if (nb > 0) hipLaunchKernelGGL(tri_scatter_kernel, frame_grid(nb, n_frames), dim3(CB), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_triangulate(hipStream_t st, const float K[9], const Pose* X_host,
                              const float* d_X16, const int32_t* d_pairs, int n, const int* d_n,
                              const float* d_p1, int n1, const float* d_p2, int n2,
                              const float* d_app2, float* d_out_xyz, int32_t* d_out_pairs,
                              float* d_out_app, int* d_n_out, int* d_scratch) {
  return launch_triangulate_batch(st, K, X_host, d_X16, d_pairs, n, d_n, d_p1, n1, d_p2, n2, d_app2, d_out_xyz,
                                  d_out_pairs, d_out_app, d_n_out, d_scratch, 1, 0, 0, 0, 0);
}

// ---- join ---------------------------------------------------------------------------
// table[ref] = (position of the FIRST world pair whose .first == ref) << 32 | that pair's .second
// (vo_complete.cpp:57-62 scans from the start and breaks on the first hit): the minimum over the positions, taken by a
// 64-bit atomicMin so that the partner's world index rides along -- the lookups then cost ONE random read per image pair
// instead of two dependent ones (table, then the world pair).
constexpr unsigned long long JOIN_EMPTY = 0x7f7f7f7f7f7f7f7full;   // byte pattern of the memset

__global__ __launch_bounds__(256) void join_build_kernel(const int32_t* __restrict__ world, int n_max,
                                                         const int* __restrict__ d_n, int n_ref,
                                                         unsigned long long* table, size_t world_stride) {
  world += 2 * blockIdx.y * world_stride;            // frame = blockIdx.y
  table += (size_t)blockIdx.y * n_ref;
  if (d_n) d_n += blockIdx.y;
  const int n = clamp_count(d_n, n_max);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int2 pr = reinterpret_cast<const int2*>(world)[j];
    if (pr.x >= 0 && pr.x < n_ref) atomicMin(&table[pr.x], ((unsigned long long)(unsigned)j << 32) | (unsigned)pr.y);
  }
}

struct JoinArgs {
  const int32_t* img; int n_max; const int* d_n;
  const int32_t* world;
  int n_ref;
  const unsigned long long* table;
  int32_t* out;
  int* counts;
  size_t img_stride, world_stride, out_stride, counts_stride;   // per frame, in pairs / ints
  int nb, n_frames;                                             // workgroups per frame, frames
  int* tmp_w;                                                   // [n_frames][n_max]: what the counting pass looked up ...
  unsigned long long* tmp_ok;                                   // [n_frames][(n_max + 63) / 64]: ... and whether there was a partner
  unsigned long long* chain;                                    // single-pass form: [n_frames][chain_stride] status words
  size_t chain_stride;
};
// the writing pass with the solver's gather folded in: BatchArgs by value beside the join's own arguments
struct JoinSink { BatchArgs b; };

__device__ __forceinline__ JoinArgs join_frame(JoinArgs a, int frame) {
  const size_t f = (size_t)frame;
  a.img += 2 * f * a.img_stride;
  if (a.d_n) a.d_n += f;
  a.world += 2 * f * a.world_stride;
  a.table += f * (size_t)a.n_ref;
  a.out += 2 * f * a.out_stride;
  a.counts += f * a.counts_stride;
  if (a.tmp_w) { a.tmp_w += f * (size_t)a.n_max; a.tmp_ok += f * (size_t)((a.n_max + 63) / 64); }
  if (a.chain) a.chain += f * a.chain_stride;
  return a;
}

__device__ __forceinline__ bool join_eval(const JoinArgs& a, int i, int& cur, int& w) {
  const int2 pr = reinterpret_cast<const int2*>(a.img)[i];
  cur = pr.y;
  if (pr.x < 0 || pr.x >= a.n_ref) return false;
  const unsigned long long e = a.table[pr.x];
  if (e == JOIN_EMPTY) return false;
  w = (int)(unsigned)(e & 0xffffffffull);
  return true;
}

__global__ __launch_bounds__(CB) void join_count_kernel(JoinArgs a0) {
  const FrameBlock fb = frame_block(a0.nb, a0.n_frames);
  if (!fb.live) return;
  const JoinArgs a = join_frame(a0, fb.f);
  __shared__ int s_wave[CB / 64];
  const int n = clamp_count(a.d_n, a.n_max);
  const int i = fb.b * CB + threadIdx.x;
  bool ok = false;
  int c, w = 0;
  if (i < n) ok = join_eval(a, i, c, w);
  // kept for the writing pass: one coalesced read there instead of the lookup again
  const unsigned long long m = __ballot(ok);
  if (i < a.n_max) {
    a.tmp_w[i] = w;
    if ((threadIdx.x & 63) == 0) a.tmp_ok[i >> 6] = m;
  }
  int total;
  block_rank(ok, s_wave, total);
  if (threadIdx.x == 0) a.counts[fb.b] = total;
}

template <bool SINK>
__device__ __forceinline__ void join_scatter_body(const JoinArgs& a0, const BatchArgs* sink) {
  const FrameBlock fb = frame_block(a0.nb, a0.n_frames);
  if (!fb.live) return;
  const JoinArgs a = join_frame(a0, fb.f);
  __shared__ int s_wave[CB / 64];
  const int i = fb.b * CB + threadIdx.x;
  bool ok = false;
  int c = 0, w = 0;
  if (i < a.n_max) {
    ok = (a.tmp_ok[i >> 6] >> (i & 63)) & 1ull;        // join_count_kernel's lookup
    if (ok) { w = a.tmp_w[i]; c = a.img[2 * (size_t)i + 1]; }
  }
  Pose Xw;
  if (SINK) Xw = batch_pack_pose(*sink, fb.f);
  int total;
  const int r = block_rank(ok, s_wave, total);
  if (ok) {
    const size_t dst = (size_t)a.counts[fb.b] + r;
    a.out[2 * dst] = c;        // vo_complete.cpp:59
    a.out[2 * dst + 1] = w;
    // the solver's gather for this pair, here where it is in registers (picp_batch_pack_kernel would read it back);
    // a pair beyond the solver's capacity is not packed, as there
    if (SINK && dst < sink->cap) batch_pack_item(*sink, fb.f, Xw, dst, c, w);
  }
}
__global__ __launch_bounds__(CB) void join_scatter_kernel(JoinArgs a0) { join_scatter_body<false>(a0, nullptr); }
__global__ __launch_bounds__(CB) void join_scatter_pack_kernel(JoinArgs a0, JoinSink s) { join_scatter_body<true>(a0, &s.b); }

// The join in ONE pass (chain_scan.h): look the partner up, rank, learn the offset from the predecessors' counts, write
// the pair -- and, with a sink, the solver's packed correspondence, whose point and pixel are requested before the
// look-back (their addresses come from the pair, not from the slot).
template <bool SINK>
__device__ __forceinline__ void join_onepass_body(const JoinArgs& a0, const BatchArgs* sink, int* d_n_out) {
  const FrameBlock fb = frame_block(a0.nb, a0.n_frames);
  if (!fb.live) return;
  const JoinArgs a = join_frame(a0, fb.f);
  __shared__ int s_cnt[OP_ITEMS * (CB / 64)];
  __shared__ int s_b, s_excl;
  if (threadIdx.x == 0) s_b = chain_take_ticket(a.chain);
  __syncthreads();
  const int b = s_b;
  const int n = clamp_count(a.d_n, a.n_max);
  bool ok[OP_ITEMS];
  int c[OP_ITEMS], w[OP_ITEMS], rank[OP_ITEMS];
#pragma unroll
  for (int j = 0; j < OP_ITEMS; ++j) {
    const int i = b * OP_CHUNK + j * CB + threadIdx.x;
    ok[j] = false; c[j] = 0; w[j] = 0;
    if (i < n) ok[j] = join_eval(a, i, c[j], w[j]);
  }
  PackedItem it[OP_ITEMS];
  if (SINK) {
    const Pose Xw = batch_pack_pose(*sink, fb.f);
#pragma unroll
    for (int j = 0; j < OP_ITEMS; ++j) {
      it[j] = PackedItem{0.f, 0.f, 0.f, 0.f, 0.f};
      if (ok[j]) it[j] = batch_pack_load(*sink, fb.f, Xw, c[j], w[j]);
    }
  }
  int total;
  chunk_rank(ok, s_cnt, rank, total);
  if (threadIdx.x < 64) {
    const int e = chain_lookback(a.chain, b, total);
    if (threadIdx.x == 0) s_excl = e;
  }
  __syncthreads();
  const int excl = s_excl;
  if (excl < 0) return;
#pragma unroll
  for (int j = 0; j < OP_ITEMS; ++j) {
    if (ok[j]) {
      const size_t dst = (size_t)excl + rank[j];
      reinterpret_cast<int2*>(a.out)[dst] = make_int2(c[j], w[j]);        // vo_complete.cpp:59
      if (SINK && dst < sink->cap) batch_pack_store(*sink, fb.f, dst, it[j]);     // a pair beyond the solver's capacity is not packed
    }
  }
  if (b == a.nb - 1 && threadIdx.x == 0) d_n_out[fb.f] = excl + total;
}
__global__ __launch_bounds__(CB) void join_onepass_kernel(JoinArgs a0, int* d_n_out) { join_onepass_body<false>(a0, nullptr, d_n_out); }
__global__ __launch_bounds__(CB) void join_onepass_pack_kernel(JoinArgs a0, JoinSink s, int* d_n_out) { join_onepass_body<true>(a0, &s.b, d_n_out); }

bool join_fuses_gather(int n_img, int n_world, int n_ref) { return !(n_img <= SMALL_N && n_ref <= SMALL_N && n_world <= 8 * SMALL_N) && n_img > 0; }

__global__ __launch_bounds__(CB) void join_small_kernel(JoinArgs a0, int n_world_max, const int* d_n_world, int* d_n_out) {
  const int f = blockIdx.x;
  const JoinArgs a = join_frame(a0, f);
  __shared__ unsigned long long s_tab[SMALL_N > 0 ? SMALL_N : 1];
  __shared__ int s_wave[CB / 64];
  for (int k = threadIdx.x; k < a.n_ref; k += CB) s_tab[k] = JOIN_EMPTY;
  __syncthreads();
  const int nw = clamp_count(d_n_world ? d_n_world + f : nullptr, n_world_max);
  for (int j = threadIdx.x; j < nw; j += CB) {           // as join_build_kernel
    const int2 pr = reinterpret_cast<const int2*>(a.world)[j];
    if (pr.x >= 0 && pr.x < a.n_ref) atomicMin(&s_tab[pr.x], ((unsigned long long)(unsigned)j << 32) | (unsigned)pr.y);
  }
  __syncthreads();
  const int n = clamp_count(a.d_n, a.n_max);
  int first = 0;
  for (int base = 0; base < n; base += CB) {
    const int i = base + threadIdx.x;
    bool ok = false;
    int c = 0, w = 0;
    if (i < n) {
      const int2 pr = reinterpret_cast<const int2*>(a.img)[i];
      c = pr.y;
      if (pr.x >= 0 && pr.x < a.n_ref) {
        const unsigned long long e = s_tab[pr.x];
        ok = e != JOIN_EMPTY;
        w = (int)(unsigned)(e & 0xffffffffull);
      }
    }
    int total;
    const int r = block_rank(ok, s_wave, total);
    if (ok) reinterpret_cast<int2*>(a.out)[(size_t)first + r] = make_int2(c, w);      // vo_complete.cpp:59
    first += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) d_n_out[f] = first;
}

// n_frames > 1: frame f joins d_img + f*img_stride (d_n_img[f] pairs) with d_world + f*world_stride
// (d_n_world[f] pairs, or n_world when null) into d_out + f*out_stride, count in d_n_out[f];
// d_table: n_frames*n_ref 64-bit words, d_scratch: n_frames*compaction_scratch_ints(n_img) ints.
hipError_t launch_join_batch(hipStream_t st, const int32_t* d_img, int n_img, const int* d_n_img,
                             const int32_t* d_world, int n_world, const int* d_n_world, int n_ref, int32_t* d_out,
                             int* d_n_out, unsigned long long* d_table, int* d_scratch, int n_frames, size_t img_stride,
                             size_t world_stride, size_t out_stride, const BatchArgs* sink) {
  const bool batched = n_frames > 1;
  hipError_t e = hipSuccess;
  if (n_img <= SMALL_N && n_ref <= SMALL_N && n_world <= 8 * SMALL_N) {
    JoinArgs a{d_img, n_img, d_n_img, d_world, n_ref, d_table, d_out, d_scratch,
               batched ? img_stride : 0, batched ? world_stride : 0, batched ? out_stride : 0, 0, 0, n_frames, nullptr, nullptr, nullptr, 0};
    hipLaunchKernelGGL(join_small_kernel, dim3(n_frames), dim3(CB), 0, st, a, n_world, d_n_world, d_n_out);
    return hipGetLastError();
  }
  if (n_ref > 0) {
    e = hipMemsetAsync(d_table, 0x7f, sizeof(unsigned long long) * (size_t)n_ref * (size_t)n_frames, st);
    if (e != hipSuccess) return e;
  }
  if (n_world > 0 && n_ref > 0) {
    int grid = (n_world + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(join_build_kernel, dim3(grid, n_frames), dim3(256), 0, st, d_world, n_world, d_n_world, n_ref,
                       d_table, batched ? world_stride : 0);
  }
  const int nb = (n_img + CB - 1) / CB;
  JoinArgs a{d_img, n_img, d_n_img, d_world, n_ref, d_table, d_out, d_scratch,
             batched ? img_stride : 0, batched ? world_stride : 0, batched ? out_stride : 0,
             batched ? compaction_scratch_ints(n_img) : 0, nb, n_frames, nullptr, nullptr,
             reinterpret_cast<unsigned long long*>(d_scratch), chain_stride(n_img)};
  if (nb == 0) return hipMemsetAsync(d_n_out, 0, sizeof(int) * (size_t)n_frames, st);
  if (!two_pass()) {
    e = hipMemsetAsync(d_scratch, 0, sizeof(int) * compaction_scratch_ints(n_img) * (size_t)n_frames, st);
    if (e != hipSuccess) return e;
    a.nb = (n_img + OP_CHUNK - 1) / OP_CHUNK;                   // chunks of OP_ITEMS rows
    if (sink) {
      JoinSink js; js.b = *sink;
      hipLaunchKernelGGL(join_onepass_pack_kernel, frame_grid(a.nb, n_frames), dim3(CB), 0, st, a, js, d_n_out);
    } else {
      hipLaunchKernelGGL(join_onepass_kernel, frame_grid(a.nb, n_frames), dim3(CB), 0, st, a, d_n_out);
    }
    return hipGetLastError();
  }
  {
    size_t off = (sizeof(int) * compaction_scratch_ints(n_img) * (size_t)n_frames + 15) & ~(size_t)15;      // join_scratch_bytes
    char* t = reinterpret_cast<char*>(d_scratch) + off;
    a.tmp_ok = reinterpret_cast<unsigned long long*>(t);
    a.tmp_w = reinterpret_cast<int*>(t + sizeof(unsigned long long) * (size_t)((n_img + 63) / 64) * (size_t)n_frames);
  }
  if (nb > 0) hipLaunchKernelGGL(join_count_kernel, frame_grid(nb, n_frames), dim3(CB), 0, st, a);
  e = launch_scan(st, d_scratch, nb, d_n_out, nullptr, n_frames, a.counts_stride);
  if (e != hipSuccess) return e;
  if (nb > 0 && sink) {
    JoinSink js; js.b = *sink;
    hipLaunchKernelGGL(join_scatter_pack_kernel, frame_grid(nb, n_frames), dim3(CB), 0, st, a, js);
  } else if (nb > 0) {
    hipLaunchKernelGGL(join_scatter_kernel, frame_grid(nb, n_frames), dim3(CB), 0, st, a);
  }
  return hipGetLastError();
}

hipError_t launch_join(hipStream_t st, const int32_t* d_img, int n_img, const int* d_n_img,
                       const int32_t* d_world, int n_world, const int* d_n_world, int n_ref,
                       int32_t* d_out, int* d_n_out, unsigned long long* d_table, int* d_scratch) {
  return launch_join_batch(st, d_img, n_img, d_n_img, d_world, n_world, d_n_world, n_ref, d_out, d_n_out, d_table,
                           d_scratch, 1, 0, 0, 0);
}

// ---- matcher output compaction (used by match.hip) -------------------------------
// best[q] = (bits(d2) << 32) | tree index, or idx 0xffffffff when no hit.
struct MatchOutArgs {
  const unsigned long long* best;
  int nq;
  int tree_is_1;    // pairs are (a1 index, a2 index): vo_complete.cpp:40-43
  int32_t* out;
  int* counts;
  size_t best_stride, out_stride, counts_stride;   // per frame (blockIdx.y)
  const int* d_n1; const int* d_n2;                // ragged frames: per-frame set sizes (null: nq / tree_is_1 for all)
  int cap1, cap2;                                  // ragged frames: capacities of the two sets (sizes are clamped to them, like the search kernels do)
  const int* unres;                                // hash-first (or null): per-frame count of queries the exact-duplicate pass left open.  A frame
                                                   //   with none has every pair at its final slot already (the lookup wrote pair q at slot q): the
                                                   //   counting pass reports full blocks without reading a key, the writing pass returns
};
extern const int SMALL_COMPACT = SMALL_N;

// the frame's query count and roles: the larger set is the tree, a1 on ties (vo_complete.cpp:15-20)
__device__ __forceinline__ void match_out_frame(const MatchOutArgs& a, int& nq, int& tree_is_1) {
  nq = a.nq; tree_is_1 = a.tree_is_1;
  if (a.d_n1) {
    int n1 = a.d_n1[blockIdx.y], n2 = a.d_n2[blockIdx.y];
    // the SAME clamp as match_kernel<RAGGED> / cell_sets: otherwise an out-of-contract size (n1 > cap1) could pick another
    // tree here than the search did, and the (ref, cur) columns would come out swapped
    n1 = n1 < 0 ? 0 : (n1 > a.cap1 ? a.cap1 : n1); n2 = n2 < 0 ? 0 : (n2 > a.cap2 ? a.cap2 : n2);
    tree_is_1 = n1 >= n2;
    const int q = tree_is_1 ? n2 : n1;
    nq = q < a.nq ? q : a.nq;
  }
}

__global__ __launch_bounds__(CB) void match_count_kernel(MatchOutArgs a) {
  __shared__ int s_wave[CB / 64];
  const unsigned long long* best = a.best + blockIdx.y * a.best_stride;
  const int q = blockIdx.x * CB + threadIdx.x;
  int nq, tree_is_1;
  match_out_frame(a, nq, tree_is_1);
  if (a.unres && a.unres[blockIdx.y] == 0) {              // every query matched: this block's count is its number of queries
    const int left = nq - (int)blockIdx.x * CB;
    if (threadIdx.x == 0) a.counts[blockIdx.y * a.counts_stride + blockIdx.x] = left < 0 ? 0 : (left > CB ? CB : left);
    return;
  }
  const bool ok = q < nq && (unsigned)(best[q] & 0xffffffffull) != 0xffffffffu;
  int total;
  block_rank(ok, s_wave, total);
  if (threadIdx.x == 0) a.counts[blockIdx.y * a.counts_stride + blockIdx.x] = total;
}

__global__ __launch_bounds__(CB) void match_scatter_kernel(MatchOutArgs a) {
  __shared__ int s_wave[CB / 64];
  const unsigned long long* best = a.best + blockIdx.y * a.best_stride;
  int32_t* out = a.out + 2 * blockIdx.y * a.out_stride;
  const int q = blockIdx.x * CB + threadIdx.x;
  if (a.unres && a.unres[blockIdx.y] == 0) return;        // the lookup wrote every pair where it belongs
  int nq, tree_is_1;
  match_out_frame(a, nq, tree_is_1);
  unsigned idx = 0xffffffffu;
  if (q < nq) idx = (unsigned)(best[q] & 0xffffffffull);
  const bool ok = idx != 0xffffffffu;
  int total;
  const int r = block_rank(ok, s_wave, total);
  if (ok) {
    const size_t dst = (size_t)a.counts[blockIdx.y * a.counts_stride + blockIdx.x] + r;
    out[2 * dst] = tree_is_1 ? (int)idx : q;
    out[2 * dst + 1] = tree_is_1 ? q : (int)idx;
  }
}

// the same in one pass (chain_scan.h); a frame whose queries all found their copy (unres == 0) has its pairs in place
__global__ __launch_bounds__(CB) void match_onepass_kernel(MatchOutArgs a, int* d_n_out, unsigned long long* chain, size_t chain_stride) {
  __shared__ int s_cnt[OP_ITEMS * (CB / 64)];
  __shared__ int s_b, s_excl;
  int nq, tree_is_1;
  match_out_frame(a, nq, tree_is_1);
  if (a.unres && a.unres[blockIdx.y] == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) d_n_out[blockIdx.y] = nq;
    return;
  }
  chain += blockIdx.y * chain_stride;
  if (threadIdx.x == 0) s_b = chain_take_ticket(chain);
  __syncthreads();
  const int b = s_b;
  const unsigned long long* best = a.best + blockIdx.y * a.best_stride;
  int32_t* out = a.out + 2 * blockIdx.y * a.out_stride;
  bool ok[OP_ITEMS];
  unsigned idx[OP_ITEMS];
  int rank[OP_ITEMS];
#pragma unroll
  for (int j = 0; j < OP_ITEMS; ++j) {
    const int q = b * OP_CHUNK + j * CB + threadIdx.x;
    idx[j] = 0xffffffffu;
    if (q < nq) idx[j] = (unsigned)(best[q] & 0xffffffffull);
    ok[j] = idx[j] != 0xffffffffu;
  }
  int total;
  chunk_rank(ok, s_cnt, rank, total);
  if (threadIdx.x < 64) {
    const int e = chain_lookback(chain, b, total);
    if (threadIdx.x == 0) s_excl = e;
  }
  __syncthreads();
  const int excl = s_excl;
  if (excl < 0) return;
#pragma unroll
  for (int j = 0; j < OP_ITEMS; ++j) {
    if (ok[j]) {
      const int q = b * OP_CHUNK + j * CB + threadIdx.x;
      const size_t dst = (size_t)excl + rank[j];
      reinterpret_cast<int2*>(out)[dst] = tree_is_1 ? make_int2((int)idx[j], q) : make_int2(q, (int)idx[j]);
    }
  }
  if (b == (int)gridDim.x - 1 && threadIdx.x == 0) d_n_out[blockIdx.y] = excl + total;
}

__global__ __launch_bounds__(CB) void match_compact_small_kernel(MatchOutArgs a, int* d_n_out) {
  __shared__ int s_wave[CB / 64];
  const unsigned long long* best = a.best + blockIdx.y * a.best_stride;
  int32_t* out = a.out + 2 * blockIdx.y * a.out_stride;
  int nq, tree_is_1;
  match_out_frame(a, nq, tree_is_1);
  int first = 0;
  for (int base = 0; base < nq; base += CB) {
    const int q = base + threadIdx.x;
    unsigned idx = 0xffffffffu;
    if (q < nq) idx = (unsigned)(best[q] & 0xffffffffull);
    const bool ok = idx != 0xffffffffu;
    int total;
    const int r = block_rank(ok, s_wave, total);
    if (ok) {
      const size_t dst = (size_t)first + r;
      out[2 * dst] = tree_is_1 ? (int)idx : q;
      out[2 * dst + 1] = tree_is_1 ? q : (int)idx;
    }
    first += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) d_n_out[blockIdx.y] = first;
}

hipError_t launch_match_compact(hipStream_t st, const unsigned long long* d_best, int nq, int tree_is_1,
                                int32_t* d_out, int* d_n_out, int* d_scratch, int n_frames, size_t best_stride,
                                size_t out_stride, const int* d_n1, const int* d_n2, int cap1, int cap2, const int* d_unres) {
  const int nb = (nq + CB - 1) / CB;
  const size_t cs = n_frames > 1 ? compaction_scratch_ints(nq) : 0;
  MatchOutArgs a{d_best, nq, tree_is_1, d_out, d_scratch, best_stride, n_frames > 1 ? out_stride : 0, cs, d_n1, d_n2, cap1, cap2, d_unres};
  if (nq <= SMALL_N) {
    hipLaunchKernelGGL(match_compact_small_kernel, dim3(1, n_frames), dim3(CB), 0, st, a, d_n_out);
    return hipGetLastError();
  }
  if (nb == 0) return hipMemsetAsync(d_n_out, 0, sizeof(int) * (size_t)n_frames, st);
  if (!two_pass()) {
    hipError_t e0 = hipMemsetAsync(d_scratch, 0, sizeof(int) * compaction_scratch_ints(nq) * (size_t)(n_frames > 1 ? n_frames : 1), st);
    if (e0 != hipSuccess) return e0;
    hipLaunchKernelGGL(match_onepass_kernel, dim3((nq + OP_CHUNK - 1) / OP_CHUNK, n_frames), dim3(CB), 0, st, a, d_n_out,
                       reinterpret_cast<unsigned long long*>(d_scratch), chain_stride(nq));
    return hipGetLastError();
  }
  if (nb > 0) hipLaunchKernelGGL(match_count_kernel, dim3(nb, n_frames), dim3(CB), 0, st, a);
  hipError_t e = launch_scan(st, d_scratch, nb, d_n_out, nullptr, n_frames, cs);
  if (e != hipSuccess) return e;
  if (nb > 0) hipLaunchKernelGGL(match_scatter_kernel, dim3(nb, n_frames), dim3(CB), 0, st, a);
  return hipGetLastError();
}

}  // namespace vo
