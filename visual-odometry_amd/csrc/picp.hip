// picp.hip -- projective-ICP Gauss-Newton kernels for gfx950 (MI355X).
//
// Replaces the scalar host loop PICPSolver::linearize / oneRound
// (picp_solver.cpp:55-112) and errorAndJacobian (:25-53):
//
//   picp_pack_kernel   once per solve: gathers world[pair.second] and
//                      meas[pair.first] into five SoA arrays (20 B per
//                      correspondence) so that every iteration streams
//                      coalesced data and never touches the index pairs again.
//   picp_round_kernel  one launch per Gauss-Newton iteration.  Every workgroup
//                      first re-derives the pose of this iteration from the
//                      previous launch's workgroup partials (fixed-order
//                      reduction + 6x6 pivoted LDLT + pose update, redundantly
//                      and identically in every workgroup), then linearises its
//                      slice and writes one 30-float partial.  One kernel
//                      boundary per iteration, no atomics, no in-launch
//                      inter-workgroup hand-off, bitwise reproducible.
//   picp_batch_*       many independent problems, one 1024-thread workgroup
//                      each, all iterations inside a single launch.
//
// Reduction: per-thread registers -> DPP row reduction (16 lanes) -> LDS across
// rows and waves -> workgroup partial.  No MFMA: the normal equations are a
// tall-skinny accumulate (21+6+3 sums per correspondence), not a contraction.
#include <type_traits>
#include <stdlib.h>

#include "vo_internal.h"
#include "chain_scan.h"

namespace vo {

// ---- wave-level helpers -----------------------------------------------------
// bound_ctrl=true with the source as "old" lets the DPP-combine pass fold the
// move into the add (v_add_f32_dpp): one instruction per reduction step.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// sum over the 16 lanes of a DPP row; every lane of the row gets the result
__device__ __forceinline__ float row16_allsum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}

// Reduce the NACC per-thread accumulators over a workgroup of NWAVES waves.
// On return threads 0..31 hold in `out` the workgroup sum of slot threadIdx.x
// (slots >= NACC are 0).  s_red: NWAVES*4*32 floats.  Fixed summation order.
template <int NWAVES>
__device__ __forceinline__ float block_reduce_acc(float acc[NACC], float* s_red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = row16_allsum(acc[k]);
  if ((lane & 15) == 15) {
    float* dst = s_red + (wave * 4 + (lane >> 4)) * 32;
#pragma unroll
    for (int k = 0; k < NACC; ++k) dst[k] = acc[k];
  }
  __syncthreads();
  float out = 0.f;
  if (tid < NACC) {
#pragma unroll 8
    for (int j = 0; j < NWAVES * 4; ++j) out += s_red[j * 32 + tid];
  }
  return out;
}

// Workgroup reduction of the round kernel (256 threads): rows of ACC_STRIDE floats in LDS (conflict-free 16-byte accesses),
// a first stage over parts of rows, a second one over the parts.  (Rounds 1-2 stored all 30 accumulators of every thread:
// git history, block_reduce_lds256.)
constexpr int ACC_STRIDE = 36;
constexpr int PICP_PARTS = PICP_BLOCK / 32;          // 32-row parts of the first reduction stage
constexpr int PICP_GROUPS = PICP_BLOCK / 8;          // row groups of the partial-row fetch (8 threads per 128-B row)
constexpr int STG_STRIDE = PICP_GROUPS + 4;          // floats per slot of the transposed staging (conflict-free 16-B reads)
constexpr int PICP_SACC = PICP_BLOCK * ACC_STRIDE > 32 * STG_STRIDE ? PICP_BLOCK * ACC_STRIDE : 32 * STG_STRIDE;
// The reduction with a quad pre-reduction in registers: two DPP adds per accumulator leave every quad's sum in its four
// lanes, lane q of the quad then stores accumulators 8q .. 8q+7 -- TWO 16-byte LDS writes per lane instead of eight (a
// ds_write_b128 costs ~13 cycles of the wave's LDS path whatever it holds, and the four waves share that path) -- and the
// first stage adds 8 rows per part instead of 32.  Rows: one per quad (PICP_BLOCK / 4), stride ACC_STRIDE.  Fixed order:
// bitwise reproducible, and the same function serves the round kernels and the one-launch small-problem kernel.
__device__ __forceinline__ float block_reduce_quad(float acc[NACC], float* s_acc, float* s_part) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    acc[k] += dpp_mov<0xB1>(acc[k]);    // quad_perm [1,0,3,2]
    acc[k] += dpp_mov<0x4E>(acc[k]);    // quad_perm [2,3,0,1]
  }
  const int q = tid & 3;
  const bool q1 = q == 1, q2 = q == 2, q3 = q == 3;
  float w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // (the operands are made opaque: the compiler turns a select chain over acc[8q + j] into an INDEXED read of acc[],
    // i.e. parks the accumulators in scratch memory)
    float a0 = acc[j], a1 = acc[8 + j], a2 = acc[16 + j], a3 = 24 + j < NACC ? acc[24 + j < NACC ? 24 + j : 0] : 0.f;
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    float v = a0;
    v = q1 ? a1 : v;
    v = q2 ? a2 : v;
    v = q3 ? a3 : v;
    w[j] = v;
  }
  float4* row = reinterpret_cast<float4*>(s_acc + (tid >> 2) * ACC_STRIDE + 8 * q);
  row[0] = make_float4(w[0], w[1], w[2], w[3]);
  row[1] = make_float4(w[4], w[5], w[6], w[7]);
  __syncthreads();
  constexpr int ROWS_PER_PART = (PICP_BLOCK / 4) / PICP_PARTS;      // 8
  const int slot = tid & 31, part = tid >> 5;
  const float* src = s_acc + (part * ROWS_PER_PART) * ACC_STRIDE + slot;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < ROWS_PER_PART; ++j) s += src[j * ACC_STRIDE];
  s_part[part * 32 + slot] = s;
  __syncthreads();
  float out = 0.f;
  if (tid < 32) {
#pragma unroll
    for (int g = 0; g < PICP_PARTS; ++g) out += s_part[g * 32 + tid];
  }
  return out;
}

__device__ __forceinline__ Pose load_pose12(const float* p) {
  Pose P;
#pragma unroll
  for (int i = 0; i < 9; ++i) P.R[i] = p[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) P.t[i] = p[9 + i];
  return P;
}

// The pose is the same in every lane: move it to scalar registers.
__device__ __forceinline__ Pose uniform_pose(const Pose& P) {
  Pose U;
#pragma unroll
  for (int i = 0; i < 9; ++i) U.R[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(P.R[i])));
#pragma unroll
  for (int i = 0; i < 3; ++i) U.t[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(P.t[i])));
  return U;
}

__device__ __forceinline__ void store_pose12(float* p, const Pose& P) {
#pragma unroll
  for (int i = 0; i < 9; ++i) p[i] = P.R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) p[9 + i] = P.t[i];
}

// ---- tail of oneRound (picp_solver.cpp:102-110) without pivoting and without LDS ------------------------------------
// (Rounds 1-2 followed Eigen's pivot order through LDS -- picp_tail_expand / picp_tail_wave, git history; the reference-order
// mode still does: picp_update_t<true>.)
// H = sum(lambda J^T J) + damping * I is symmetric positive definite by construction, and an LDL^T of such a matrix is
// backward stable in ANY pivot order (no element growth: Higham, Accuracy and Stability, thm 10.3) -- Eigen pivots because
// its LDLT also serves semidefinite and indefinite matrices.  The fast mode therefore eliminates in the natural order: no
// ranking, no permutation through LDS (two dependent round trips), no un-permutation.  The lower triangle and -b are taken
// straight out of the lanes that hold them (27 v_readlane, compile-time lane numbers); factorisation and substitutions are
// ldlt6_solve_ordered, redundant in all lanes; the three angles go through the small-angle sin / cos in three lanes.
// Against the reference's arithmetic this is one more rounding-level deviation of the fast mode (vo_math.h), the
// reference-order mode keeps Eigen's pivoted factorisation.
// The pose composition T <- v2t(dx) * T (picp_solver.cpp:110) is done ACROSS lanes: lane l < 12 computes entry l of the new
// pose (l = r + 3c: R(r,c) for c < 3, t(r) for c = 3) as one 3-term product a . b with a = row r of dR (three selects from
// the uniform dR) and b = column c of the old pose, which the lane holds in (b0, b1, b2) -- the round kernel loads them
// straight from the previous launch's pose (one 12-byte load per lane at kernel start), the in-launch forms select them from
// their uniform copy (pose_lane_operands).  Same products, same order (dot3, + dt for the translation) as pose_mul: the same
// bits, in ~20 instead of ~63 instructions of every lane.
__device__ __forceinline__ void pose_lane_operands(const Pose& T, float& b0, float& b1, float& b2) {
  // plain scalars and one select per step: a chain of selects over the members of T makes the compiler park T in scratch
  // memory and select among ADDRESSES (the trap of DESIGN.md section 4.2)
  const int lane = threadIdx.x & 63;
  const float r0 = T.R[0], r1 = T.R[1], r2 = T.R[2], r3 = T.R[3], r4 = T.R[4], r5 = T.R[5], r6 = T.R[6], r7 = T.R[7], r8 = T.R[8];
  const float t0 = T.t[0], t1 = T.t[1], t2 = T.t[2];
  const bool c1 = lane >= 3, c2 = lane >= 6, c3 = lane >= 9;
  b0 = r0; b1 = r1; b2 = r2;
  b0 = c1 ? r3 : b0; b1 = c1 ? r4 : b1; b2 = c1 ? r5 : b2;
  b0 = c2 ? r6 : b0; b1 = c2 ? r7 : b1; b2 = c2 ? r8 : b2;
  b0 = c3 ? t0 : b0; b1 = c3 ? t1 : b1; b2 = c3 ? t2 : b2;
}

__device__ __forceinline__ Pose picp_tail_direct(float val, float b0, float b1, float b2) {
  const int lane = threadIdx.x & 63;
  float B[6][6], y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j)
      B[i][j] = j <= i ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), 6 * i + j)) : 0.f;
    y[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), 36 + i));
  }
  ldlt6_solve_ordered(B, y);
  const int m = lane % 3;
  const float y3 = y[3], y4 = y[4], y5 = y[5];
  float ang = y3;
  ang = m == 1 ? y4 : ang;
  ang = m == 2 ? y5 : ang;
  float sn, cs;
  if (__builtin_expect(!(fabsf(y3) <= 0.5f && fabsf(y4) <= 0.5f && fabsf(y5) <= 0.5f), 0)) sincosf(ang, &sn, &cs);
  else sincos_small(ang, sn, cs);
  const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), 0));
  const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), 0));
  const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), 1));
  const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), 1));
  const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), 2));
  const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), 2));
  const Pose dT = v2t_from_sincos(y, sx, cx, sy, cy, sz, cz);
  // entry l = r + 3c of v2t(dx) * T in lane l < 12
  const float d0 = dT.R[0], d1 = dT.R[1], d2 = dT.R[2], d3 = dT.R[3], d4 = dT.R[4], d5 = dT.R[5], d6 = dT.R[6], d7 = dT.R[7], d8 = dT.R[8];
  const float u0 = dT.t[0], u1 = dT.t[1], u2 = dT.t[2];
  const bool m1 = m == 1, m2 = m == 2;
  float a0 = d0, a1 = d3, a2 = d6, dt = u0;
  a0 = m1 ? d1 : a0; a1 = m1 ? d4 : a1; a2 = m1 ? d7 : a2; dt = m1 ? u1 : dt;
  a0 = m2 ? d2 : a0; a1 = m2 ? d5 : a1; a2 = m2 ? d8 : a2; dt = m2 ? u2 : dt;
  float e = dot3(a0, b0, a1, b1, a2, b2);
  e = lane >= 9 ? e + dt : e;
  Pose Tn;
#pragma unroll
  for (int i = 0; i < 9; ++i) Tn.R[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), i));
#pragma unroll
  for (int i = 0; i < 3; ++i) Tn.t[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 9 + i));
  return Tn;
}

// what lane l of the solving wave contributes: H(l / 6, l % 6) + damping on the diagonal from the reduced accumulators
// (s_tot: NACC floats in LDS), -b(l - 36), 0 beyond.  Optionally publishes H (col-major) and b.
__device__ __forceinline__ float picp_lane_value(const float* s_tot, float damping, float* H_out, float* b_out) {
  const int lane = threadIdx.x & 63;
  float val = 0.f;
  if (lane < 36) {
    const int r = (lane * 43) >> 8, c = lane - 6 * r;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    val = s_tot[(13 * lo - lo * lo) / 2 + (hi - lo)];      // row-major upper triangle
    if (r == c) val += 1.f * damping;                      // picp_solver.cpp:102
    if (H_out) H_out[r + 6 * c] = val;
  } else if (lane < 42) {
    const float bv = s_tot[21 + (lane - 36)];
    val = -bv;                                             // picp_solver.cpp:109 solve(-b)
    if (b_out) b_out[lane - 36] = bv;
  }
  return val;
}

// ---- pack -------------------------------------------------------------------
__global__ __launch_bounds__(256) void picp_pack_kernel(const int32_t* __restrict__ pairs,
                                                        const int* __restrict__ d_n, int n_max,
                                                        const float* __restrict__ world, int n_world,
                                                        const float* __restrict__ meas, int n_meas,
                                                        PackedCorr pk, PicpParams* P, PicpState* S,
                                                        const float* __restrict__ T0) {
  int n = n_max;
  if (d_n) { const int m = *d_n; n = m < n_max ? (m < 0 ? 0 : m) : n_max; }
  if (blockIdx.x == 0 && threadIdx.x == 0) P->n_corr = n;
  if (T0 && blockIdx.x == 0 && threadIdx.x < 12) {          // pending pose reset rides along
    const int k = threadIdx.x;
    S->pose[0][k] = k < 9 ? T0[(k % 3) + 4 * (k / 3)] : T0[12 + (k - 9)];
  }
  const float qnan = __int_as_float((int)VO_DROPPED_BITS);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int m = pairs[2 * i];       // .first  -> measurement (picp_solver.cpp:66)
    const int w = pairs[2 * i + 1];   // .second -> world point (picp_solver.cpp:67)
    float x = qnan, y = 0.f, z = 0.f, u = 0.f, v = 0.f;
    if (m >= 0 && m < n_meas && w >= 0 && w < n_world) {
      x = world[3 * (size_t)w]; y = world[3 * (size_t)w + 1]; z = world[3 * (size_t)w + 2];
      u = meas[2 * (size_t)m]; v = meas[2 * (size_t)m + 1];
    } else {
      atomicAdd(&S->n_bad, 1);
    }
    pk.arr(0)[i] = x; pk.arr(1)[i] = y; pk.arr(2)[i] = z; pk.arr(3)[i] = u; pk.arr(4)[i] = v;
  }
}

#ifdef VO_STAMPS
#define VO_STAMP(k)                                                                       \
  do {                                                                                    \
    if (blockIdx.x == 0 && threadIdx.x == 0 && it < 128) {                                \
      unsigned long long _t;                                                              \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");         \
      S->stamps[it][k] = _t;                                                              \
    }                                                                                     \
  } while (0)
// the constant 100 MHz reference clock beside the shader-clock stamp: calibrates s_memtime's tick per step
#define VO_STAMP_REAL(k)                                                                  \
  do {                                                                                    \
    if (blockIdx.x == 0 && threadIdx.x == 0 && it < 128) {                                \
      unsigned long long _t;                                                              \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");     \
      S->stamps[it][k] = _t;                                                              \
    }                                                                                     \
  } while (0)
#else
#define VO_STAMP(k) do {} while (0)
#define VO_STAMP_REAL(k) do {} while (0)
#endif

// ---- one Gauss-Newton round ---------------------------------------------------
// PRE:    first derive this round's pose from the partials of launch it-1.
// FINISH: only derive the pose (single workgroup), publish the statistics.
// BATCH: several problems per launch, problem = blockIdx.y (a few problems with many workgroups each; many problems
// go through picp_batch_kernel instead).  Everything per-problem is reached through strides.
struct RoundBatch {
  const int* n_pairs;        // live correspondences per problem
  size_t cap;                // capacity of a problem's packed arrays
  size_t partials_stride;    // floats between two problems' partial buffers
  float* T_out;              // FINISH: n_problems x 16
  float* stats_out;          // FINISH: n_problems x 4, or null
};

template <bool PRE, bool FINISH, bool PINHOLE, bool KEEP, bool BATCH>
__device__ __forceinline__ void picp_round_body(const PicpParams* __restrict__ P, PicpState* S, PackedCorr pk,
                                                float* partials, int it, int nb, const RoundBatch& rb) {
  if (BATCH) {
    const size_t p = blockIdx.y;
    S += p;
    pk.base += p * 5 * rb.cap;
    partials += p * rb.partials_stride;
  }
  __shared__ __attribute__((aligned(16))) float s_acc[PICP_SACC];   // also the staging of the partial rows
  __shared__ float s_part[PICP_PARTS * 32];
  __shared__ float s_stat[4];
  const int tid = threadIdx.x;
  VO_STAMP(0);
  VO_STAMP_REAL(7);

  // (1) The previous launch's workgroup partials (nb rows of 32 floats, zero-padded to
  // a multiple of 256 rows) depend on kernel arguments only: their loads go out first,
  // before anything that waits on a parameter load.  Thread (g = tid/8, q = tid%8) owns
  // the 16-B quad q of rows g + 32j: eight independent 16-B loads per pass (the rows were
  // written by workgroups on all eight XCDs: Infinity-Cache/HBM round trips that must
  // overlap, not chain).
  float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
  // The previous round's pose (written by workgroup 0 of the previous launch, i.e. on another XCD for most readers) goes out
  // with the partial rows, as VECTOR loads: lane l fetches the column of the old pose it will multiply by in the tail
  // (picp_tail_direct: entries 3c .. 3c+2 of the 12 floats, c = l / 3 clamped to 3).
  float pb0 = 0.f, pb1 = 0.f, pb2 = 0.f;
  if (PRE) {
    const int l = tid & 63;
    const int c3 = l < 3 ? 0 : (l < 6 ? 3 : (l < 9 ? 6 : 9));
    const float* pp = S->pose[(it - 1) & (PICP_SLOTS - 1)] + c3;
    pb0 = pp[0]; pb1 = pp[1]; pb2 = pp[2];
  }
  if (PRE) {
    const int nb_pad = (nb + 255) & ~255;
    const float* prev = partials + (size_t)(blockIdx.x % PICP_REPLICAS) * PICP_SLOTS * nb_pad * PICP_PSTRIDE + (size_t)((it - 1) & (PICP_SLOTS - 1)) * nb_pad * PICP_PSTRIDE;
    const float4* src = reinterpret_cast<const float4*>(prev + (size_t)(tid >> 3) * PICP_PSTRIDE) + (tid & 7);
    constexpr int NJ = 256 / PICP_GROUPS;              // loads per thread and pass of 256 rows
    for (int b0 = 0; b0 < nb; b0 += 256) {
      float4 r[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        r[j] = src[(size_t)(b0 + PICP_GROUPS * j) * (PICP_PSTRIDE / 4)];      // (rows >= nb read as zero)
#pragma unroll
      for (int j = 0; j < NJ; ++j) { psum.x += r[j].x; psum.y += r[j].y; psum.z += r[j].z; psum.w += r[j].w; }
    }
  }

  // (2) this thread's first correspondence (coalesced SoA loads), in flight during the solve
  int n = P->n_corr;
  if (BATCH) {
    n = rb.n_pairs[blockIdx.y];
    n = n < 0 ? 0 : ((size_t)n > rb.cap ? (int)rb.cap : n);
  }
  // camera and threshold: fetched here, while the partial rows are in flight, and pinned in scalar registers (left
  // alone the compiler re-loads them after the solve, on the chain)
  CamK cam = P->cam;
  float thr = P->thr;
  float damping = P->damping;
  asm volatile("" : "+s"(damping), "+s"(nb));
  if (!FINISH) {
#pragma unroll
    for (int k = 0; k < 9; ++k) asm volatile("" : "+s"(cam.K[k]));
    asm volatile("" : "+s"(cam.rows), "+s"(cam.cols), "+s"(cam.z_near), "+s"(cam.z_far), "+s"(thr));
  }
  int i = blockIdx.x * PICP_BLOCK + tid;
  bool have = !FINISH && i < n;
  float x = 0.f, y = 0.f, z = 0.f, u = 0.f, v = 0.f;
  if (have) { x = pk.arr(0)[i]; y = pk.arr(1)[i]; z = pk.arr(2)[i]; u = pk.arr(3)[i]; v = pk.arr(4)[i]; }

  Pose T;
  if (PRE) {
    VO_STAMP(1);
    // (3) stage the 32 group sums transposed: s_acc[slot*36 + group], so that one slot's 32
    // values are eight conflict-free 16-B reads.
    {
      const int g32 = tid >> 3, q4 = (tid & 7) * 4;
      s_acc[(q4 + 0) * STG_STRIDE + g32] = psum.x;
      s_acc[(q4 + 1) * STG_STRIDE + g32] = psum.y;
      s_acc[(q4 + 2) * STG_STRIDE + g32] = psum.z;
      s_acc[(q4 + 3) * STG_STRIDE + g32] = psum.w;
    }
    __syncthreads();
    // (4) Every wave finishes the sums it needs and runs the (uniform) 6x6 solve on its own:
    // the four waves sit on four SIMDs, so the redundancy is free and spares three
    // workgroup barriers plus the LDS broadcast of the pose.  Lane l < 36 builds H(r,c)
    // (+damping on the diagonal, picp_solver.cpp:102), lanes 36..41 -b, lanes 42..44 the
    // statistics.  Fixed order => every wave of every workgroup gets the same bits.
    const int wave = tid >> 6, lane = tid & 63;
    float val = 0.f;                                         // this lane's entry of the system (picp_tail_direct)
    if (lane < 45) {
      int slot;
      bool diag = false;
      if (lane < 36) {
        const int r = lane / 6, c = lane - 6 * r;
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        slot = (13 * lo - lo * lo) / 2 + (hi - lo);          // row-major upper triangle
        diag = r == c;
      } else {
        slot = 21 + (lane - 36);                             // 21..26 b, 27..29 chi_in, chi_out, n_in
      }
      const float4* row = reinterpret_cast<const float4*>(s_acc + slot * STG_STRIDE);
      float tsum = 0.f;
#pragma unroll
      for (int k = 0; k < PICP_GROUPS / 4; ++k) { const float4 t4 = row[k]; tsum += t4.x; tsum += t4.y; tsum += t4.z; tsum += t4.w; }
      if (lane < 36) {
        val = diag ? tsum + 1.f * damping : tsum;
        if (FINISH && blockIdx.x == 0 && wave == 0) S->H[(lane % 6) * 6 + lane / 6] = val;    // col-major
      } else if (lane < 42) {
        val = -tsum;                                         // picp_solver.cpp:109 solve(-b)
        if (FINISH && blockIdx.x == 0 && wave == 0) S->b[lane - 36] = tsum;
      } else if (wave == 0) {
        s_stat[lane - 42] = tsum;
      }
    }
    VO_STAMP(2);
    {
      const Pose Tn = picp_tail_direct(val, pb0, pb1, pb2);
      if (tid == 0 && blockIdx.x == 0) {
        store_pose12(S->pose[FINISH ? 0 : (it & (PICP_SLOTS - 1))], Tn);
        if (FINISH) {
          float T16[16];
          pose_to_T16(Tn, T16);
#pragma unroll
          for (int k = 0; k < 16; ++k) S->T16[k] = T16[k];
          S->chi_in = s_stat[0];
          S->chi_out = s_stat[1];
          S->n_in = (int)(s_stat[2] + 0.5f);
          if (BATCH) {
#pragma unroll
            for (int k = 0; k < 16; ++k) rb.T_out[16 * (size_t)blockIdx.y + k] = T16[k];
            if (rb.stats_out) {
              float* so = rb.stats_out + 4 * (size_t)blockIdx.y;
              so[0] = s_stat[0]; so[1] = s_stat[1]; so[2] = s_stat[2]; so[3] = 0.f;
            }
          }
        }
      }
      T = uniform_pose(Tn);
    }
    VO_STAMP(3);
    if (FINISH) return;
    __syncthreads();   // every wave is done with the staged rows: s_acc is reused by the reduction
  } else {
    T = uniform_pose(load_pose12(S->pose[0]));
  }

  float acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
  const int stride = gridDim.x * PICP_BLOCK;
  while (have) {
    const float cx = x, cy = y, cz = z, cu = u, cv = v;
    i += stride;
    have = i < n;
    if (have) { x = pk.arr(0)[i]; y = pk.arr(1)[i]; z = pk.arr(2)[i]; u = pk.arr(3)[i]; v = pk.arr(4)[i]; }
    picp_accumulate_t<PINHOLE, KEEP>(cam, T, thr, cx, cy, cz, cu, cv, acc);
  }
  VO_STAMP(4);
  const float tot = block_reduce_quad(acc, s_acc, s_part);
  VO_STAMP(5);
  if (tid < PICP_PSTRIDE) {
    float o = tot;   // slot 29 (inlier count) is an exact integer in float: < 2^24 correspondences
    if (tid >= NACC) o = 0.f;
#pragma unroll
    for (int r = 0; r < PICP_REPLICAS; ++r)
      partials[(size_t)r * PICP_SLOTS * ((nb + 255) & ~255) * PICP_PSTRIDE + ((size_t)(it & (PICP_SLOTS - 1)) * ((nb + 255) & ~255) + blockIdx.x) * PICP_PSTRIDE + tid] = o;
  }
  VO_STAMP(6);
}

// single problem: no batch view among the kernel arguments (a launch graph of 101 of these is replayed per frame)
// Its arguments are all scalars (no aggregate), so that -amdgpu-kernarg-preload-count hands every one of them to
// the wave in SGPRs at launch: no kernel-argument load at the head of the chain.
template <bool PRE, bool FINISH, bool PINHOLE, bool KEEP>
__global__ __launch_bounds__(PICP_BLOCK) void picp_round_kernel(const PicpParams* __restrict__ P, PicpState* S,
                                                                float* pk_base, size_t pk_cap, float* partials, int it,
                                                                int nb) {
  picp_round_body<PRE, FINISH, PINHOLE, KEEP, false>(P, S, PackedCorr{pk_base, pk_cap}, partials, it, nb, RoundBatch{});
}

// a few problems per launch: problem = blockIdx.y
template <bool PRE, bool FINISH, bool PINHOLE, bool KEEP>
__global__ __launch_bounds__(PICP_BLOCK) void picp_round_batch_kernel(const PicpParams* __restrict__ P, PicpState* S,
                                                                      PackedCorr pk, float* partials, int it, int nb,
                                                                      RoundBatch rb) {
  picp_round_body<PRE, FINISH, PINHOLE, KEEP, true>(P, S, pk, partials, it, nb, rb);
}

int picp_grid_for(int n_corr, int n_cu) {
  // correspondences per thread (VO_PICP_PER_THREAD in the environment, default 1): fewer workgroups mean fewer partial rows
  // for every workgroup of the next round to wait for, against a longer linearisation in each
  static const int per_thread = [] { const char* e = getenv("VO_PICP_PER_THREAD"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 16 ? v : 1; }();
  int g = (n_corr + PICP_BLOCK * per_thread - 1) / (PICP_BLOCK * per_thread);
  if (g < 1) g = 1;
  // beyond one workgroup per CU let each thread take several correspondences
  // before adding workgroups: the per-iteration partial reduction reads
  // grid*128 B in every workgroup.
  const int cap = n_cu > 0 ? 4 * n_cu : 1024;
  if (g > cap) g = cap;
  if (g > PICP_MAX_BLOCKS) g = PICP_MAX_BLOCKS;
  return g;
}

hipError_t launch_picp_pack(hipStream_t st, const int32_t* d_pairs, const int* d_n, int n_max,
                            const float* d_world, int n_world, const float* d_meas, int n_meas,
                            PackedCorr pk, PicpParams* d_params, PicpState* d_state, const float* d_T0) {
  hipError_t e = hipMemsetAsync(&d_state->n_bad, 0, sizeof(int), st);
  if (e != hipSuccess) return e;
  int grid = (n_max + 255) / 256;
  if (grid < 1) grid = 1;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(picp_pack_kernel, dim3(grid), dim3(256), 0, st, d_pairs, d_n, n_max, d_world,
                     n_world, d_meas, n_meas, pk, d_params, d_state, d_T0);
  return hipGetLastError();
}

// ---- all rounds of ONE small problem in one launch ------------------------------------------------
// A problem of at most PICP_BLOCK correspondences is one workgroup in the launch-per-round form, whose round then costs a kernel
// boundary and a trip of its single partial row through memory for nothing (5.3 us, whatever the size -- the reference's dataset has
// <= 127 points per frame).  Here the same workgroup keeps going: its correspondence stays in registers, the reduced row goes
// through LDS, and the arithmetic is EXACTLY that of the launch-per-round form with one workgroup (same accumulation, same
// workgroup reduction; the sum over "all partial rows" there is 0 + row + 0 + ..., i.e. the row, a -0.0f turned into +0.0f),
// so the two forms return the same bits (tests/test_gpu_more.py::test_small_problem_form_equals_the_round_kernels).
template <bool PINHOLE, bool KEEP>
__global__ __launch_bounds__(PICP_BLOCK) void picp_small_kernel(const PicpParams* __restrict__ P, PicpState* S, const float* pk_base,
                                                                size_t pk_cap, int n_iters) {
  __shared__ __attribute__((aligned(16))) float s_acc[PICP_SACC];
  __shared__ float s_part[PICP_PARTS * 32];
  __shared__ float s_tot[32];
  __shared__ float s_stat[4];
  const PackedCorr pk{const_cast<float*>(pk_base), pk_cap};
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int n = P->n_corr;
  if (n > PICP_BLOCK) n = PICP_BLOCK;                   // (the host only picks this form for at most PICP_BLOCK correspondences)
  const CamK cam = P->cam;
  const float thr = P->thr, damping = P->damping;
  const bool have = tid < n;
  float x = 0.f, y = 0.f, z = 0.f, u = 0.f, v = 0.f;
  if (have) { x = pk.arr(0)[tid]; y = pk.arr(1)[tid]; z = pk.arr(2)[tid]; u = pk.arr(3)[tid]; v = pk.arr(4)[tid]; }
  Pose T = uniform_pose(load_pose12(S->pose[0]));
  for (int it = 0; it < n_iters; ++it) {
    const bool last = it == n_iters - 1;
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
    if (have) picp_accumulate_t<PINHOLE, KEEP>(cam, T, thr, x, y, z, u, v, acc);
    const float tot = block_reduce_quad(acc, s_acc, s_part);
    if (tid < 32) s_tot[tid] = tid < NACC ? tot : 0.f;
    __syncthreads();
    // every wave builds the system and solves it on its own (as step (4) of picp_round_body)
    float val = 0.f;
    if (lane < 45) {
      int slot;
      bool diag = false;
      if (lane < 36) {
        const int r = lane / 6, c = lane - 6 * r;
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        slot = (13 * lo - lo * lo) / 2 + (hi - lo);
        diag = r == c;
      } else {
        slot = 21 + (lane - 36);
      }
      const float tsum = 0.f + s_tot[slot];              // what summing the one partial row and its zero padding gives
      if (lane < 36) {
        val = diag ? tsum + 1.f * damping : tsum;
        if (last && wave == 0) S->H[(lane % 6) * 6 + lane / 6] = val;    // col-major
      } else if (lane < 42) {
        val = -tsum;
        if (last && wave == 0) S->b[lane - 36] = tsum;
      } else if (wave == 0) {
        s_stat[lane - 42] = tsum;
      }
    }
    float b0, b1, b2;
    pose_lane_operands(T, b0, b1, b2);
    const Pose Tn = picp_tail_direct(val, b0, b1, b2);
    if (last && tid == 0) {
      store_pose12(S->pose[0], Tn);
      float T16[16];
      pose_to_T16(Tn, T16);
#pragma unroll
      for (int k = 0; k < 16; ++k) S->T16[k] = T16[k];
      S->chi_in = s_stat[0];
      S->chi_out = s_stat[1];
      S->n_in = (int)(s_stat[2] + 0.5f);
    }
    T = uniform_pose(Tn);
    __syncthreads();                                     // s_acc / s_tot are reused by the next round
  }
}

// VO_PICP_SMALL=0 in the environment keeps small problems on the launch-per-round form (the test of the equality above)
static bool picp_small_enabled() {
  static const bool on = [] { const char* e = getenv("VO_PICP_SMALL"); return !(e && e[0] == '0'); }();
  return on;
}

template <bool PINHOLE, bool KEEP>
static void launch_rounds_t(hipStream_t st, const PicpParams* d_params, PicpState* d_state, PackedCorr pk,
                            float* d_partials, int grid, int n_iters) {
  hipLaunchKernelGGL((picp_round_kernel<false, false, PINHOLE, KEEP>), dim3(grid), dim3(PICP_BLOCK), 0, st,
                     d_params, d_state, pk.base, pk.cap, d_partials, 0, grid);
  for (int it = 1; it < n_iters; ++it)
    hipLaunchKernelGGL((picp_round_kernel<true, false, PINHOLE, KEEP>), dim3(grid), dim3(PICP_BLOCK), 0, st,
                       d_params, d_state, pk.base, pk.cap, d_partials, it, grid);
  hipLaunchKernelGGL((picp_round_kernel<true, true, false, false>), dim3(1), dim3(PICP_BLOCK), 0, st, d_params,
                     d_state, pk.base, pk.cap, d_partials, n_iters, grid);
}

// ---- a chain of rounds without its finishing launch (vo_picp_one_round, capi.hip) ------------------------------------
// The reference's own loop (vo_complete.cpp:163-164) calls oneRound a hundred times and reads the camera afterwards.  Round
// `it` of a chain is ONE launch: it derives its pose from the partial rows of round it - 1 (it = 0: from the finished state)
// exactly as the rounds inside launch_picp_rounds do; the FINISH launch -- the last solve, H, b, statistics, T16 -- is
// enqueued by launch_picp_finish when somebody needs the state.  Same kernels, same order: the same bits as one
// launch_picp_rounds call of as many rounds.
bool picp_rounds_chain(int grid) { return !(grid == 1 && picp_small_enabled()); }

hipError_t launch_picp_chain_round(hipStream_t st, const PicpParams* d_params, PicpState* d_state, PackedCorr pk,
                                   float* d_partials, int grid, int it, bool pinhole, bool keep_outliers) {
  const dim3 g(grid), b(PICP_BLOCK);
#define VO_CHAIN_LAUNCH(PRE, PH, KP) \
  hipLaunchKernelGGL((picp_round_kernel<PRE, false, PH, KP>), g, b, 0, st, d_params, d_state, pk.base, pk.cap, d_partials, it, grid)
  if (it > 0) {
    if (pinhole) { if (keep_outliers) VO_CHAIN_LAUNCH(true, true, true); else VO_CHAIN_LAUNCH(true, true, false); }
    else { if (keep_outliers) VO_CHAIN_LAUNCH(true, false, true); else VO_CHAIN_LAUNCH(true, false, false); }
  } else {
    if (pinhole) { if (keep_outliers) VO_CHAIN_LAUNCH(false, true, true); else VO_CHAIN_LAUNCH(false, true, false); }
    else { if (keep_outliers) VO_CHAIN_LAUNCH(false, false, true); else VO_CHAIN_LAUNCH(false, false, false); }
  }
#undef VO_CHAIN_LAUNCH
  return hipGetLastError();
}

// closes a chain of `n_rounds` rounds (the state is then what launch_picp_rounds(n_rounds) leaves)
hipError_t launch_picp_finish(hipStream_t st, const PicpParams* d_params, PicpState* d_state, PackedCorr pk,
                              float* d_partials, int grid, int n_rounds) {
  hipLaunchKernelGGL((picp_round_kernel<true, true, false, false>), dim3(1), dim3(PICP_BLOCK), 0, st, d_params, d_state,
                     pk.base, pk.cap, d_partials, n_rounds, grid);
  return hipGetLastError();
}

hipError_t launch_picp_rounds(hipStream_t st, const PicpParams* d_params, PicpState* d_state,
                              PackedCorr pk, float* d_partials, int grid, int n_iters, bool pinhole,
                              bool keep_outliers) {
  if (n_iters <= 0) return hipSuccess;
  if (grid == 1 && picp_small_enabled()) {      // one workgroup's worth of correspondences (picp_grid_for): all rounds in one launch
    const dim3 g(1), b(PICP_BLOCK);
    if (pinhole && keep_outliers) hipLaunchKernelGGL((picp_small_kernel<true, true>), g, b, 0, st, d_params, d_state, pk.base, pk.cap, n_iters);
    else if (pinhole) hipLaunchKernelGGL((picp_small_kernel<true, false>), g, b, 0, st, d_params, d_state, pk.base, pk.cap, n_iters);
    else if (keep_outliers) hipLaunchKernelGGL((picp_small_kernel<false, true>), g, b, 0, st, d_params, d_state, pk.base, pk.cap, n_iters);
    else hipLaunchKernelGGL((picp_small_kernel<false, false>), g, b, 0, st, d_params, d_state, pk.base, pk.cap, n_iters);
    return hipGetLastError();
  }
  if (pinhole) {
    if (keep_outliers) launch_rounds_t<true, true>(st, d_params, d_state, pk, d_partials, grid, n_iters);
    else launch_rounds_t<true, false>(st, d_params, d_state, pk, d_partials, grid, n_iters);
  } else {
    if (keep_outliers) launch_rounds_t<false, true>(st, d_params, d_state, pk, d_partials, grid, n_iters);
    else launch_rounds_t<false, false>(st, d_params, d_state, pk, d_partials, grid, n_iters);
  }
  return hipGetLastError();
}

// ---- batched solver -----------------------------------------------------------
__global__ __launch_bounds__(256) void picp_batch_pack_kernel(BatchArgs a) {
  const FrameBlock fb = frame_block(a.pack_gx, a.n_problems);
  if (!fb.live) return;
  const int p = fb.f;
  int n = a.n_pairs[p];
  if (n < 0) n = 0;
  if ((size_t)n > a.cap) n = (int)a.cap;
  const int32_t* pairs = a.pairs + 2 * (size_t)p * a.pairs_stride;
  if (a.states && fb.b == 0 && threadIdx.x < 12) {       // launch-per-round form: the problem's starting pose
    const int k = threadIdx.x;
    float v;
    if (a.T0) v = k < 9 ? a.T0[16 * (size_t)p + (k % 3) + 4 * (k / 3)] : a.T0[16 * (size_t)p + 12 + (k - 9)];
    else v = (k < 9 && (k % 4) == 0) ? 1.f : 0.f;
    a.states[p].pose[0][k] = v;
  }
  const Pose Xw = batch_pack_pose(a, p);
  for (int i = fb.b * 256 + threadIdx.x; i < n; i += fb.nb * 256) batch_pack_item(a, p, Xw, (size_t)i, pairs[2 * i], pairs[2 * i + 1]);
}

// after the solver: the number of correspondences dropped for an index outside the point arrays, per problem
__global__ void picp_batch_bad_kernel(BatchArgs a) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < a.n_problems) a.stats_out[4 * (size_t)p + 3] = (float)a.n_bad[p];
}

// zero rounds in the launch-per-round form: T_out = T0 (or identity), statistics zero
__global__ void picp_batch_T0_out_kernel(BatchArgs a) {
  const size_t p = blockIdx.x;
  const int k = threadIdx.x;
  if (k < 16) a.T_out[16 * p + k] = a.T0 ? a.T0[16 * p + k] : ((k % 5) == 0 ? 1.f : 0.f);
  if (k < 4 && a.stats_out) a.stats_out[4 * p + k] = 0.f;
}

// The rotation lives in scalar registers; the translation is kept in three VECTOR registers: pc = fma(R, w, t) may read one
// scalar operand only (constant bus), so a scalar t would cost a v_mov per component and correspondence in a loop that
// is bound by VALU issue.
#define VO_BATCH_T_VGPR(T) asm volatile("" : "+v"((T).t[0]), "+v"((T).t[1]), "+v"((T).t[2]))

template <bool PINHOLE, bool KEEP>
__global__ __launch_bounds__(PICP_BATCH_BLOCK) void picp_batch_kernel(BatchArgs a) {
  __shared__ float s_red[(PICP_BATCH_BLOCK / 64) * 4 * 32];
  __shared__ float s_tot[32];
  __shared__ float s_pose[12];
  const int tid = threadIdx.x;
  const int p = blockIdx.x;
  int n = a.n_pairs[p];
  if (n < 0) n = 0;
  if ((size_t)n > a.cap) n = (int)a.cap;
  const float* X = a.packed + (size_t)p * 5 * a.cap;
  const float* Y = X + a.cap;
  const float* Z = Y + a.cap;
  const float* U = Z + a.cap;
  const float* V = U + a.cap;
  Pose T;
  if (a.T0) {
    T = pose_from_T16(a.T0 + 16 * (size_t)p);
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) T.R[k] = (k % 4 == 0) ? 1.f : 0.f;
    T.t[0] = T.t[1] = T.t[2] = 0.f;
  }
  T = uniform_pose(T);
  VO_BATCH_T_VGPR(T);
  const CamK cam = a.cam;
  const int n4 = n & ~3;
  // The workgroup owns its CU (168 VGPRs x 768 threads), so the CU's LDS is otherwise idle: the first
  // PICP_BATCH_LDS_TRIPS trips of every thread (its own float4 groups, written in round 0, read back by
  // the same thread in every later round -- no barrier, conflict-free b128) never touch HBM again.
  __shared__ float4 s_cache[PICP_BATCH_LDS_TRIPS][5][PICP_BATCH_BLOCK];
  // One round.  The statistics (chi of inliers / outliers, inlier count: 6 of ~110 vector instructions per correspondence
  // in a loop bound by VALU issue) are what the LAST round reports (picp_solver.h:44-50 read after oneRound); the rounds
  // before it run the instantiation without them.
  auto round = [&](int it, auto stats_tag) {
    constexpr bool STATS = decltype(stats_tag)::value;
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
    // the first streamed trip is requested before the LDS-resident trips are linearised: all waves of the workgroup start a
    // round together (barrier), so without this they would all sit in that first memory wait together
    int i = tid * 4 + PICP_BATCH_LDS_TRIPS * PICP_BATCH_BLOCK * 4;
    bool have = i < n4;
    float4 x, y, z, u, v;
    // UNCONDITIONAL loads, here and in the loop below (a trip beyond the problem re-reads a valid one and is not used): under a
    // predicate the compiler cannot count what is in flight, and waits for the NEXT trip's loads before it finishes the
    // current one (s_waitcnt vmcnt(4) ... vmcnt(0) right behind their issue): the double buffering then hides nothing inside
    // a wave.  With three waves per SIMD the other two cover most of it: 1.61 -> 1.56 ms per 200 x 50k x 50 rounds on one box,
    // 1.57 -> 1.59 on another -- inside the box-to-box spread; kept for the wait counts.  (A ring of three buffers, or the same
    // two as arrays indexed by a constant, spill at the 170 registers three waves per SIMD leave: 1.74 / 1.60 ms.)
    {
      const int ic = have ? i : 0;
      x = *reinterpret_cast<const float4*>(X + ic); y = *reinterpret_cast<const float4*>(Y + ic);
      z = *reinterpret_cast<const float4*>(Z + ic); u = *reinterpret_cast<const float4*>(U + ic);
      v = *reinterpret_cast<const float4*>(V + ic);
    }
#pragma unroll
    for (int c = 0; c < PICP_BATCH_LDS_TRIPS; ++c) {
      const int j = tid * 4 + c * PICP_BATCH_BLOCK * 4;
      if (j < n4) {
        float4 cx, cy, cz, cu, cv;
        if (it == 0) {
          cx = *reinterpret_cast<const float4*>(X + j); cy = *reinterpret_cast<const float4*>(Y + j);
          cz = *reinterpret_cast<const float4*>(Z + j); cu = *reinterpret_cast<const float4*>(U + j);
          cv = *reinterpret_cast<const float4*>(V + j);
          s_cache[c][0][tid] = cx; s_cache[c][1][tid] = cy; s_cache[c][2][tid] = cz;
          s_cache[c][3][tid] = cu; s_cache[c][4][tid] = cv;
        } else {
          cx = s_cache[c][0][tid]; cy = s_cache[c][1][tid]; cz = s_cache[c][2][tid];
          cu = s_cache[c][3][tid]; cv = s_cache[c][4][tid];
        }
        picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, cx.x, cy.x, cz.x, cu.x, cv.x, acc);
        picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, cx.y, cy.y, cz.y, cu.y, cv.y, acc);
        picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, cx.z, cy.z, cz.z, cu.z, cv.z, acc);
        picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, cx.w, cy.w, cz.w, cu.w, cv.w, acc);
      }
    }
    // register double buffering: the next trip's five 16-B loads are in flight while the current four correspondences are
    // linearised.  Two trips per pass of the loop, the buffers swapping roles, so that no register is copied from "next"
    // to "current" (20 v_mov per trip otherwise, 4 % of a VALU-bound loop).
    float4 x2, y2, z2, u2, v2;
    while (have) {
      i += PICP_BATCH_BLOCK * 4;
      bool have2 = i < n4;
      {
        const int ic = have2 ? i : i - PICP_BATCH_BLOCK * 4;      // (the trip being linearised: valid)
        x2 = *reinterpret_cast<const float4*>(X + ic); y2 = *reinterpret_cast<const float4*>(Y + ic);
        z2 = *reinterpret_cast<const float4*>(Z + ic); u2 = *reinterpret_cast<const float4*>(U + ic);
        v2 = *reinterpret_cast<const float4*>(V + ic);
      }
      picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, x.x, y.x, z.x, u.x, v.x, acc);
      picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, x.y, y.y, z.y, u.y, v.y, acc);
      picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, x.z, y.z, z.z, u.z, v.z, acc);
      picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, x.w, y.w, z.w, u.w, v.w, acc);
      if (!have2) break;
      i += PICP_BATCH_BLOCK * 4;
      have = i < n4;
      {
        const int ic = have ? i : i - PICP_BATCH_BLOCK * 4;
        x = *reinterpret_cast<const float4*>(X + ic); y = *reinterpret_cast<const float4*>(Y + ic);
        z = *reinterpret_cast<const float4*>(Z + ic); u = *reinterpret_cast<const float4*>(U + ic);
        v = *reinterpret_cast<const float4*>(V + ic);
      }
      picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, x2.x, y2.x, z2.x, u2.x, v2.x, acc);
      picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, x2.y, y2.y, z2.y, u2.y, v2.y, acc);
      picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, x2.z, y2.z, z2.z, u2.z, v2.z, acc);
      picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, x2.w, y2.w, z2.w, u2.w, v2.w, acc);
    }
    for (int i = n4 + tid; i < n; i += PICP_BATCH_BLOCK) {
      picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, X[i], Y[i], Z[i], U[i], V[i], acc);
    }
    const float tot = block_reduce_acc<PICP_BATCH_BLOCK / 64>(acc, s_red);
    if (tid < 32) s_tot[tid] = tot;
    __syncthreads();
    if (tid < 64) {
      float b0, b1, b2;
      pose_lane_operands(T, b0, b1, b2);
      const Pose Tn = picp_tail_direct(picp_lane_value(s_tot, a.damping, nullptr, nullptr), b0, b1, b2);
      if (tid == 0) {
        store_pose12(s_pose, Tn);
        if (it == a.n_iters - 1 && a.stats_out) {
          float* so = a.stats_out + 4 * (size_t)p;
          so[0] = s_tot[27]; so[1] = s_tot[28]; so[2] = s_tot[29]; so[3] = 0.f;
        }
      }
    }
    __syncthreads();
    T = uniform_pose(load_pose12(s_pose));
    VO_BATCH_T_VGPR(T);
  };
  for (int it = 0; it < a.n_iters; ++it) {
    if (it == a.n_iters - 1) round(it, std::true_type{});
    else round(it, std::false_type{});
  }
  if (tid == 0) {
    float T16[16];
    pose_to_T16(T, T16);
#pragma unroll
    for (int k = 0; k < 16; ++k) a.T_out[16 * (size_t)p + k] = T16[k];
    if (a.n_iters <= 0 && a.stats_out) {                   // no round: no statistics
      float* so = a.stats_out + 4 * (size_t)p;
      so[0] = so[1] = so[2] = so[3] = 0.f;
    }
  }
}

// ---- batched solver, fewer problems than CUs: the waves of the idle CUs take work off the problems' own workgroups -----
// picp_batch_kernel is bound by VALU issue on the CUs it holds, and a call of 200 problems holds 200 of 256.  Here the launch
// has one workgroup per CU: workgroup p < n_problems is problem p's HOME (poses, tail, results: everything picp_batch_kernel
// does), the waves of the others are HELPERS, each on its own.  A home keeps the first `keep` trips of its problem (a trip:
// PICP_BATCH_BLOCK x 4 correspondences, one float4 group per thread); what lies beyond is cut into CHUNKS of `C` wave-trips
// (a wave-trip: 64 x 4 correspondences), and chunk g of the launch, counted in problem order, belongs to helper wave g.
// Per round a helper wave waits for the pose of its chunk's problem, linearises the chunk, reduces it over its lanes and
// publishes the 30 sums; the home adds the sums of its chunks IN CHUNK ORDER to its own, solves, and publishes the next
// pose.  A helper wave meets no barrier and no other wave: twelve chunks per CU are in flight independently, which is what
// hides the hand-over (an earlier form with one SEGMENT per helper workgroup at a time -- pose, data, reduction, barriers in
// series, 3.6 segments per round -- left the homes waiting: no gain at 200 problems).
//
// What travels between workgroups are 8-byte words (tag << 32 | float bits), tag = the round the value belongs to, read
// and written with agent-scope relaxed atomics (the accesses that go past a CU's L1 and an XCD's L2; a round trip between
// two workgroups is 1.0-1.3 us, tools/micro/hop_latency.hip): a value and its validity arrive together, no fence, single
// buffered (a home cannot publish pose r + 1 before every helper has delivered round r, i.e. has read pose r; a helper
// cannot deliver round r + 1 before it has read pose r + 1, i.e. before the home has taken round r).
//
// The result does not depend on which wave computed a chunk, and therefore not on timing: a chunk's sums are a function of
// (problem, pose, range) computed by ONE wave in a fixed order, the chunks are canonical (keep and C are functions of the
// launch: problem count and sizes), the order of the additions is fixed.  That is also what makes every wait BOUNDED: a
// home that has polled HELP_POLLS_HOME times for a chunk has one of its own waves compute it -- the same instructions on
// the same data, the same bits -- and never waits for that chunk again; a helper that finds a pose tagged beyond its round
// (the home has gone on without it), or none within HELP_POLLS_HELPER polls, leaves.  Nothing assumes that the workgroups of
// the launch are resident together (another stream's kernel may hold CUs): late helpers cost time, not correctness, and
// every wave reaches its exit.  Two problems with the same data in one launch get the same bits (same keep, same C).
constexpr int HELP_TRIP = PICP_BATCH_BLOCK * 4;
constexpr int HELP_WAVES = PICP_BATCH_BLOCK / 64;
constexpr int HELP_MAXP = 1024;                 // problems the partition tables hold (the form serves n_problems < CUs)
constexpr int HELP_MAXCHUNK = 64;               // chunks per problem (a home's set of chunks it computes itself is one 64-bit word)
constexpr int HELP_SLACK10 = 20;                // what a helper's round costs beyond its wave-trips, in tenths of a trip
constexpr int HELP_ITER100 = 71;                // a wave-trip on a CU full of helper waves, in hundredths of a home's trip
constexpr unsigned HELP_POLLS_HOME = 1u << 9;   // ~0.6 us each: a few hundred microseconds
constexpr unsigned HELP_POLLS_HELPER = 1u << 13;

// wave-trips per chunk for a home that keeps `keep` trips: what a helper wave finishes within the home's round
__device__ __forceinline__ int help_chunk_len(int keep, int slack10) {
  const int c = ((keep * 10 - slack10) * 10) / HELP_ITER100;
  return c < 1 ? 1 : c;
}

// Positions are BYTE offsets into the five arrays, 32 bits wide: a load is then scalar base + vector offset, one VGPR for all
// five arrays (64-bit addresses per array and thread -- ten VGPRs -- get hoisted out of the round loop and spilled: four
// dependent scratch reloads at the head of every round).
__device__ __forceinline__ float4 ld16(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
// float4 groups i, i + STEP, ... < end of one problem, register double buffered as in picp_batch_kernel; `between` runs
// after the first group's loads have been issued
template <bool PINHOLE, bool KEEP, bool STATS, unsigned STEP, class Between>
__device__ __forceinline__ void batch_stream(const CamK& cam, const Pose& T, float thr, const float* X, const float* Y,
                                             const float* Z, const float* U, const float* V, unsigned i, unsigned end,
                                             float acc[NACC], Between between) {
  bool have = i < end;
  float4 x, y, z, u, v;
  {
    const unsigned ic = have ? i : 0u;
    x = ld16(X, ic); y = ld16(Y, ic); z = ld16(Z, ic); u = ld16(U, ic); v = ld16(V, ic);
  }
  between();
  float4 x2, y2, z2, u2, v2;
  while (have) {
    i += STEP;
    const bool have2 = i < end;
    {
      const unsigned ic = have2 ? i : i - STEP;
      x2 = ld16(X, ic); y2 = ld16(Y, ic); z2 = ld16(Z, ic); u2 = ld16(U, ic); v2 = ld16(V, ic);
    }
    picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, thr, x.x, y.x, z.x, u.x, v.x, acc);
    picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, thr, x.y, y.y, z.y, u.y, v.y, acc);
    picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, thr, x.z, y.z, z.z, u.z, v.z, acc);
    picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, thr, x.w, y.w, z.w, u.w, v.w, acc);
    if (!have2) break;
    i += STEP;
    have = i < end;
    {
      const unsigned ic = have ? i : i - STEP;
      x = ld16(X, ic); y = ld16(Y, ic); z = ld16(Z, ic); u = ld16(U, ic); v = ld16(V, ic);
    }
    picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, thr, x2.x, y2.x, z2.x, u2.x, v2.x, acc);
    picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, thr, x2.y, y2.y, z2.y, u2.y, v2.y, acc);
    picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, thr, x2.z, y2.z, z2.z, u2.z, v2.z, acc);
    picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, thr, x2.w, y2.w, z2.w, u2.w, v2.w, acc);
  }
}

template <bool PINHOLE, bool KEEP>
__global__ __launch_bounds__(PICP_BATCH_BLOCK) void picp_batch_shared_kernel(BatchArgs a) {
  __shared__ float s_red[HELP_WAVES * 4 * 32];
  __shared__ float s_tot[32];
  __shared__ float s_pose[12];
  __shared__ float s_wave[HELP_WAVES][32];        // a wave's own 30 sums on their way from registers to lanes
  __shared__ float s_rows[HELP_MAXCHUNK][32];     // a home's chunks of the round
  __shared__ unsigned s_late[2];                  // chunks that did not arrive
  __shared__ int s_misc[8];
  __shared__ int s_n4[HELP_MAXP];                 // correspondences per problem (whole float4 groups)
  __shared__ int s_chunkpre[HELP_MAXP + 1];       // chunks before problem p
  __shared__ float4 s_cache[PICP_BATCH_LDS_TRIPS][5][PICP_BATCH_BLOCK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = a.n_problems, H = (int)gridDim.x - P;
  const CamK cam = a.cam;
  chain_word* rows = chain_ptr(a.help_words);
  chain_word* posew = rows + (size_t)a.help_rows * 32;
  auto clamp_n = [&](int p) { int n = a.n_pairs[p]; if (n < 0) n = 0; if ((size_t)n > a.cap) n = (int)a.cap; return n; };
  auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };      // (values every lane read from the same LDS word)

  // ---- the partition: every workgroup derives the same one from the problem sizes ----
  if (tid < 8) s_misc[tid] = tid == 1 ? 0x7fffffff : 0;
  __syncthreads();
  {
    int nmax = 0;
    for (int p = tid; p < P; p += PICP_BATCH_BLOCK) {
      const int n4 = clamp_n(p) & ~3;
      s_n4[p] = n4;
      nmax = n4 > nmax ? n4 : nmax;
    }
    if (nmax > 0) atomicMax(&s_misc[0], nmax);
  }
  __syncthreads();
  const int c0 = PICP_BATCH_LDS_TRIPS > 1 ? PICP_BATCH_LDS_TRIPS : 1;     // a home keeps at least what it caches in LDS
  const int Kmax_raw = (sgpr(s_misc[0]) + HELP_TRIP - 1) / HELP_TRIP;
  const int Kmax = Kmax_raw > c0 ? Kmax_raw : c0;
  const int slack10 = a.help_slack10 > 0 ? a.help_slack10 : HELP_SLACK10;
  const int budget = H * HELP_WAVES < a.help_rows ? H * HELP_WAVES : a.help_rows;      // helper waves = rows
  // chunks of problem p when the homes keep c trips (and whether any problem has more than a home can track)
  auto chunks_of = [&](int n4, int c, int C) { const int left = n4 - c * HELP_TRIP; return left > 0 ? ((left + 255) / 256 + C - 1) / C : 0; };
  if (a.help_keep > 0) {
    if (tid == 0) s_misc[1] = a.help_keep < c0 ? c0 : (a.help_keep > Kmax ? Kmax : a.help_keep);
  } else {
    // the smallest `keep` whose leftovers, cut into chunks a helper wave finishes within the home's round, are at most as
    // many as there are helper waves.  keep = Kmax (nothing left over) always passes.  One candidate per wave at a time (at
    // most 128 candidates, evenly spaced, when the problems are longer than that many trips).
    const int step = (Kmax - c0) / 128 + 1;
    for (int c = c0 + wave * step; c <= Kmax; c += HELP_WAVES * step) {
      const int Cc = help_chunk_len(c, slack10);
      int cnt = 0, most = 0;
      for (int p = lane; p < P; p += 64) { const int k = chunks_of(s_n4[p], c, Cc); cnt += k; most = k > most ? k : most; }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { cnt += __shfl_xor(cnt, d); const int o = __shfl_xor(most, d); most = o > most ? o : most; }
      if (lane == 0 && cnt <= budget && most <= HELP_MAXCHUNK) atomicMin(&s_misc[1], c);
    }
  }
  __syncthreads();
  int keep = sgpr(s_misc[1] < Kmax ? s_misc[1] : Kmax);
  int C = a.help_g > 0 ? a.help_g : help_chunk_len(keep, slack10);
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tid < 64) {                               // exclusive prefix of the chunk counts, 64 problems per step
      int carry = 0, most = 0;
      for (int base = 0; base < P; base += 64) {
        const int p = base + lane;
        const int k = p < P ? chunks_of(s_n4[p], keep, C) : 0;
        most = k > most ? k : most;
        int ki = k;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(ki, d);
          if (lane >= d) ki += o;
        }
        if (p < P) s_chunkpre[p] = carry + ki - k;
        carry += __shfl(ki, 63);
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(most, d); most = o > most ? o : most; }
      if (lane == 0) { s_chunkpre[P] = carry; s_misc[5] = most; }
    }
    __syncthreads();
    if (sgpr(s_chunkpre[P]) <= budget && sgpr(s_misc[5]) <= HELP_MAXCHUNK) break;
    keep = Kmax; C = help_chunk_len(keep, slack10);        // (pinned values that do not fit: no helper work)
    __syncthreads();
  }
  const int N = sgpr(s_chunkpre[P]);

  // One chunk: float4 groups [first, first + 256 * count) bytes-wise of problem p under pose T, by the calling WAVE alone.
  // Lanes < 32 return the chunk's sums (lane k: accumulator k, 0 beyond NACC).
  auto chunk_sums = [&](int p, int c, const Pose& T, auto stats_tag) {
    constexpr bool STATS = decltype(stats_tag)::value;
    const float* X = a.packed + (size_t)p * 5 * a.cap;
    const int n4 = clamp_n(p) & ~3;
    const unsigned first = (unsigned)(keep * HELP_TRIP + c * C * 256) * 4u;
    unsigned end = first + (unsigned)C * 1024u;
    if (end > (unsigned)n4 * 4u) end = (unsigned)n4 * 4u;
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
    batch_stream<PINHOLE, KEEP, STATS, 1024u>(cam, T, a.thr, X, X + a.cap, X + 2 * a.cap, X + 3 * a.cap, X + 4 * a.cap,
                                              first + (unsigned)lane * 16u, end, acc, [] {});
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
      float v = row16_allsum(acc[k]);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane == 0) s_wave[wave][k] = v;
    }
    return lane < NACC ? s_wave[wave][lane] : 0.f;
  };

  if ((int)blockIdx.x >= P) {
    // ---- a helper: twelve waves, each with its own chunk ----
    const int g = ((int)blockIdx.x - P) * HELP_WAVES + wave;
    if (g >= N || a.n_iters <= 0 || a.help_absent) return;
    int lo = 0, hi = P;                                     // the problem whose chunks include g
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sgpr(s_chunkpre[mid]) <= g) lo = mid; else hi = mid; }
    const int p = lo, c = g - sgpr(s_chunkpre[p]);
    for (int r = 0; r < a.n_iters; ++r) {
      float val = 0.f;
      bool ok = true;
      if (r == 0) {
        if (lane < 12) {
          if (a.T0) val = lane < 9 ? a.T0[16 * (size_t)p + (lane % 3) + 4 * (lane / 3)] : a.T0[16 * (size_t)p + 12 + (lane - 9)];
          else val = (lane < 9 && (lane % 4) == 0) ? 1.f : 0.f;
        }
      } else {
        ok = false;
        for (unsigned polls = 0; polls < HELP_POLLS_HELPER; ++polls) {
          unsigned long long w = (unsigned long long)(unsigned)r << 32;
          if (lane < 12) w = __hip_atomic_load(posew + (size_t)p * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int tag = (int)(w >> 32);
          if (__all(tag == r)) { ok = true; val = __int_as_float((int)(unsigned)w); break; }
          if (__any(tag > r)) break;                        // the home has gone on without this wave
          __builtin_amdgcn_s_sleep(2);
        }
      }
      if (!ok) return;
      Pose T;
#pragma unroll
      for (int i = 0; i < 9; ++i) T.R[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), i));
#pragma unroll
      for (int i = 0; i < 3; ++i) T.t[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(val), 9 + i));
      VO_BATCH_T_VGPR(T);
      float tot;
      if (r == a.n_iters - 1) tot = chunk_sums(p, c, T, std::true_type{});
      else tot = chunk_sums(p, c, T, std::false_type{});
      if (lane < 32)
        __hip_atomic_store(rows + (size_t)g * 32 + lane, ((unsigned long long)(unsigned)(r + 1) << 32) | (unsigned)__float_as_int(tot),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }

  // ---- a home ----
  const int p = blockIdx.x;
  const int n = clamp_n(p);
  const float* X = a.packed + (size_t)p * 5 * a.cap;
  const float* Y = X + a.cap;
  const float* Z = Y + a.cap;
  const float* U = Z + a.cap;
  const float* V = U + a.cap;
  Pose T;
  if (a.T0) {
    T = pose_from_T16(a.T0 + 16 * (size_t)p);
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) T.R[k] = (k % 4 == 0) ? 1.f : 0.f;
    T.t[0] = T.t[1] = T.t[2] = 0.f;
  }
  T = uniform_pose(T);
  VO_BATCH_T_VGPR(T);
  const int n4 = n & ~3;
  const int own_end = keep * HELP_TRIP < n4 ? keep * HELP_TRIP : n4;
  const int row0 = sgpr(s_chunkpre[p]);
  const int nchunk = sgpr(s_chunkpre[p + 1]) - row0;
  unsigned long long own = 0;                                // chunks this home computes itself from now on
  auto round = [&](int it, auto stats_tag) {
    constexpr bool STATS = decltype(stats_tag)::value;
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
    unsigned t16 = (unsigned)tid * 16u;                      // (opaque: this thread's byte offset is re-derived every round, not
    asm volatile("" : "+v"(t16));                            //  kept -- and spilled -- as ready-made addresses across the rounds)
    batch_stream<PINHOLE, KEEP, STATS, HELP_TRIP * 4u>(cam, T, a.thr, X, Y, Z, U, V, t16 + PICP_BATCH_LDS_TRIPS * HELP_TRIP * 4u,
                                                       (unsigned)own_end * 4u, acc, [&] {
#pragma unroll
      for (int c = 0; c < PICP_BATCH_LDS_TRIPS; ++c) {
        const unsigned j = t16 + c * HELP_TRIP * 4u;
        if (j < (unsigned)n4 * 4u) {
          float4 cx, cy, cz, cu, cv;
          if (it == 0) {
            cx = ld16(X, j); cy = ld16(Y, j); cz = ld16(Z, j); cu = ld16(U, j); cv = ld16(V, j);
            s_cache[c][0][tid] = cx; s_cache[c][1][tid] = cy; s_cache[c][2][tid] = cz;
            s_cache[c][3][tid] = cu; s_cache[c][4][tid] = cv;
          } else {
            cx = s_cache[c][0][tid]; cy = s_cache[c][1][tid]; cz = s_cache[c][2][tid];
            cu = s_cache[c][3][tid]; cv = s_cache[c][4][tid];
          }
          picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, cx.x, cy.x, cz.x, cu.x, cv.x, acc);
          picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, cx.y, cy.y, cz.y, cu.y, cv.y, acc);
          picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, cx.z, cy.z, cz.z, cu.z, cv.z, acc);
          picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, cx.w, cy.w, cz.w, cu.w, cv.w, acc);
        }
      }
    });
    for (int i = n4 + tid; i < n; i += PICP_BATCH_BLOCK)
      picp_accumulate_t<PINHOLE, KEEP, STATS, true>(cam, T, a.thr, X[i], Y[i], Z[i], U[i], V[i], acc);
    if (tid < 2) s_late[tid] = 0;
    float tot = block_reduce_acc<HELP_WAVES>(acc, s_red);    // (its barrier also publishes s_late = 0)
    if (nchunk > 0) {
      // the chunks of the round: every wave polls for two rows at a time (lanes 0-31 one, 32-63 the other)
      for (int j = 2 * wave + (lane >> 5); j - (lane >> 5) < nchunk; j += 2 * HELP_WAVES) {
        const bool live = j < nchunk && !((own >> j) & 1ull);
        bool got = !live;
        float val = 0.f;
        for (unsigned polls = 0; polls < HELP_POLLS_HOME; ++polls) {
          if (live && !got) {
            const unsigned long long w = __hip_atomic_load(rows + (size_t)(row0 + j) * 32 + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (a row is complete when all its 32 words carry the round's tag: checked per half wave below)
            const bool hit = (int)(w >> 32) == it + 1;
            const unsigned long long hits = __ballot(hit);
            const unsigned half = (lane >> 5) ? (unsigned)(hits >> 32) : (unsigned)hits;
            if (half == 0xffffffffu) { got = true; val = __int_as_float((int)(unsigned)w); }
          }
          if (__all(got)) break;
          __builtin_amdgcn_s_sleep(2);
        }
        if (live) {
          if (got) s_rows[j][lane & 31] = val;
          else if ((lane & 31) == 0) atomicOr(&s_late[j >> 5], 1u << (j & 31));
        }
      }
      __syncthreads();
      own |= (unsigned long long)(unsigned)sgpr((int)s_late[0]) | ((unsigned long long)(unsigned)sgpr((int)s_late[1]) << 32);
      if (own) {                                             // chunks without a helper: wave j % 12 of the home stands in
        for (int j = wave; j < nchunk; j += HELP_WAVES)
          if ((own >> j) & 1ull) {
            const float v = chunk_sums(p, j, T, stats_tag);
            if (lane < 32) s_rows[j][lane] = v;
          }
        __syncthreads();
      }
      if (tid < 32)
        for (int j = 0; j < nchunk; ++j) tot += s_rows[j][tid];    // in chunk order
    }
    if (tid < 32) s_tot[tid] = tot;
    __syncthreads();
    if (tid < 64) {
      float b0, b1, b2;
      pose_lane_operands(T, b0, b1, b2);
      const Pose Tn = picp_tail_direct(picp_lane_value(s_tot, a.damping, nullptr, nullptr), b0, b1, b2);
      if (tid == 0) {
        store_pose12(s_pose, Tn);
        if (it == a.n_iters - 1 && a.stats_out) {
          float* so = a.stats_out + 4 * (size_t)p;
          so[0] = s_tot[27]; so[1] = s_tot[28]; so[2] = s_tot[29]; so[3] = 0.f;
        }
      }
    }
    __syncthreads();
    if (nchunk > 0 && tid < 12)                              // the pose of round it + 1 (after the last round: lets late helpers go)
      __hip_atomic_store(posew + (size_t)p * 16 + tid, ((unsigned long long)(unsigned)(it + 1) << 32) | (unsigned)__float_as_int(s_pose[tid]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    T = uniform_pose(load_pose12(s_pose));
    VO_BATCH_T_VGPR(T);
  };
  for (int it = 0; it < a.n_iters; ++it) {
    if (it == a.n_iters - 1) round(it, std::true_type{});
    else round(it, std::false_type{});
  }
  if (tid == 0) {
    float T16[16];
    pose_to_T16(T, T16);
#pragma unroll
    for (int k = 0; k < 16; ++k) a.T_out[16 * (size_t)p + k] = T16[k];
    if (a.n_iters <= 0 && a.stats_out) {
      float* so = a.stats_out + 4 * (size_t)p;
      so[0] = so[1] = so[2] = so[3] = 0.f;
    }
  }
}

// ---- reference-order ("exact") solver ---------------------------------------------------
// Bit-identical to the reference's scalar float32 arithmetic (picp_solver.cpp:55-112; checked against the
// CPU restatement by tests/test_gpu_exact.py): the per-correspondence terms are computed in parallel with every
// product unfused (picp_term_exact), staged through LDS, and lane k of the first wave adds entry
// k of H / b / chi SEQUENTIALLY IN CORRESPONDENCE ORDER -- parallel across the 30 entries, serial
// across the correspondences, which is the reference's summation order.  The serial chain (one dependent
// float add per correspondence) is the floor of a round; everything else is taken off it: six producer waves
// linearise chunk c+1 into one LDS buffer while the summing wave drains chunk c from the other (one
// ds_read_b128 per four adds, no flags: an entry the reference does not add is staged as +0.0f, and
// x + (+0.0f) == x bit for bit for every x but -0.0f, which a sum started at +0.0f never holds).  The tail is Eigen's
// pivoted LDLT with true divisions (ldlt6_solve) and sin/cos in double rounded to float
// (v2t_euler_exact).  One workgroup per problem, all rounds inside the launch.  Meant for
// verification and for small frames (the reference's dataset has <= 127 points per frame, where
// a round costs a few microseconds); at 50k correspondences a round is 0.18 ms -- 50 000 dependent adds at 7.5 cycles
// each are 0.156 ms that no arrangement can remove.
constexpr int EX_BLOCK = 512;             // 8 waves: wave 0 sums, waves 1-3 and 5-7 linearise; wave 4 idles so that SIMD 0 is the summing wave's alone
constexpr int EX_CHUNK = 384;             // correspondences per chunk = producer threads
constexpr int EX_ROW = EX_CHUNK + 4;      // floats per LDS row (one row per accumulator, 16-byte aligned; +4: rows start on different banks)
static_assert(EX_CHUNK % 64 == 0, "the drain works in steps of 64 slots");

struct ExactArgs {
  const float* packed;       // 5 SoA arrays of `cap` floats per problem
  size_t cap;
  const int* n_pairs;        // per problem (device), or null: params->n_corr
  const PicpParams* params;  // single problem: camera / threshold / damping / count in device memory
  CamK cam; float thr, damping; int keep_outliers;   // batched: by value
  int n_iters;
  PicpState* state;          // single problem: pose in, everything out
  const float* T0;           // batched: n_problems x 16 or null (identity)
  float* T_out;              // batched: n_problems x 16
  float* stats_out;          // batched: n_problems x 4 or null
};

template <bool SINGLE>
__global__ __launch_bounds__(EX_BLOCK) void picp_exact_kernel(ExactArgs a) {
  __shared__ __attribute__((aligned(16))) float s_term[2][NACC * EX_ROW + 32];   // [buffer][accumulator][correspondence of the chunk] + read-ahead pad
  __shared__ float s_sum[NACC + 2];
  __shared__ float s_pose[12];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const bool producer = wave != 0 && wave != 4;
  const int pw = wave < 4 ? wave - 1 : wave - 2;                            // producer wave 0 .. 5 (waves 0 and 4 produce nothing)
  const int pt = (pw < 0 ? 0 : pw) * 64 + (tid & 63);                       // producer thread 0 .. EX_CHUNK-1
  const size_t p = blockIdx.x;
  CamK cam; float thr, damping; int keep, n;
  Pose T;
  if (SINGLE) {
    cam = a.params->cam; thr = a.params->thr; damping = a.params->damping; keep = a.params->keep_outliers;
    n = a.params->n_corr;
    T = load_pose12(a.state->pose[0]);
  } else {
    cam = a.cam; thr = a.thr; damping = a.damping; keep = a.keep_outliers;
    n = a.n_pairs[p];
    if (a.T0) {
      T = pose_from_T16(a.T0 + 16 * p);
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) T.R[k] = (k % 4 == 0) ? 1.f : 0.f;
      T.t[0] = T.t[1] = T.t[2] = 0.f;
    }
  }
  if (n < 0) n = 0;
  if ((size_t)n > a.cap) n = (int)a.cap;
  const float* X = a.packed + p * 5 * a.cap;
  const float* Y = X + a.cap;
  const float* Z = Y + a.cap;
  const float* U = Z + a.cap;
  const float* V = U + a.cap;
  const int n_chunks = (n + EX_CHUNK - 1) / EX_CHUNK;
  // what one producer thread stages for correspondence c * EX_CHUNK + pt: the value each accumulator ADDS (:72-94).
  // Its five inputs were requested one chunk ahead (ld), so that their latency hides behind the previous chunk's work.
  float px = 0.f, py = 0.f, pz = 0.f, pu = 0.f, pv = 0.f;
  auto ld = [&](int c) {
    const int i = c * EX_CHUNK + pt;
    if (i < n) { px = X[i]; py = Y[i]; pz = Z[i]; pu = U[i]; pv = V[i]; }
  };
  auto fill = [&](int c) {
    float* buf = s_term[c & 1];
    const int i = c * EX_CHUNK + pt;
    const float wx = px, wy = py, wz = pz, zu = pu, zv = pv;
    ld(c + 1 < n_chunks ? c + 1 : 0);                                     // next chunk (after the last one: the next round's first)
    float term[NTERM];
    int flag = 0;
    if (i < n) flag = picp_term_exact(cam, T, thr, wx, wy, wz, zu, zv, term);
    const bool inl = flag == 1, outl = flag == 2;
    const bool in_sys = inl || (outl && keep != 0);                       // :90-94
#pragma unroll
    for (int k = 0; k < 27; ++k) buf[k * EX_ROW + pt] = in_sys ? term[k] : 0.f;
    buf[27 * EX_ROW + pt] = inl ? term[27] : 0.f;                         // chi of the inliers (:86)
    buf[28 * EX_ROW + pt] = outl ? term[27] : 0.f;                        // chi of the outliers (:82)
    buf[29 * EX_ROW + pt] = inl ? 1.f : 0.f;                              // inlier count (:87, exact below 2^24)
  };
  if (producer && n_chunks > 0) ld(0);
  for (int it = 0; it < a.n_iters; ++it) {
    float run = 0.f;                                   // lane k < NACC of wave 0: running sum of entry k (:57-61)
    if (producer && n_chunks > 0) fill(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
      if (producer && c + 1 < n_chunks) fill(c + 1);
      if (tid < NACC) {
        // Every slot of the chunk is staged (a slot without correspondence, or whose term the reference does not add, holds
        // +0.0f), so the drain needs no bounds inside a step: steps of 64 adds, two register sets of eight float4 swapping
        // roles, each read issued 32 dependent adds (~240 cycles) before its use.  The read-ahead of the last step runs into
        // the next row / the pad behind the buffer and is never added.
        const int left = n - c * EX_CHUNK;
        const int steps = ((left < EX_CHUNK ? left : EX_CHUNK) + 63) >> 6;
        const float4* row = reinterpret_cast<const float4*>(s_term[c & 1] + tid * EX_ROW);
#define VO_RD8(q, at) { q##0 = row[at]; q##1 = row[(at) + 1]; q##2 = row[(at) + 2]; q##3 = row[(at) + 3]; \
                        q##4 = row[(at) + 4]; q##5 = row[(at) + 5]; q##6 = row[(at) + 6]; q##7 = row[(at) + 7]; }
#define VO_ADD4(v) run += v.x; run += v.y; run += v.z; run += v.w
#define VO_ADD32(q) VO_ADD4(q##0); VO_ADD4(q##1); VO_ADD4(q##2); VO_ADD4(q##3); VO_ADD4(q##4); VO_ADD4(q##5); VO_ADD4(q##6); VO_ADD4(q##7)
        float4 P0, P1, P2, P3, P4, P5, P6, P7, Q0, Q1, Q2, Q3, Q4, Q5, Q6, Q7;
        VO_RD8(P, 0)
        for (int st = 0; st < steps; ++st) {
          const int g = st * 16;
          VO_RD8(Q, g + 8)
          __builtin_amdgcn_sched_barrier(0);              // keep the issue order: the compiler would hoist every read to the top
          VO_ADD32(P);
          __builtin_amdgcn_sched_barrier(0);
          VO_RD8(P, g + 16)
          __builtin_amdgcn_sched_barrier(0);
          VO_ADD32(Q);
          __builtin_amdgcn_sched_barrier(0);
        }
#undef VO_RD8
#undef VO_ADD4
#undef VO_ADD32
      }
      __syncthreads();
    }
    if (tid < NACC) s_sum[tid] = run;
    __syncthreads();
    if (tid < 64) {                                    // one wave, every lane the same values: nothing diverges
      float acc[NACC];
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = s_sum[k];
      float H[36], b[6];
      const Pose Tn = picp_update_t<true>(acc, damping, T, H, b);          // :102-110
      if (tid == 0) {
        store_pose12(s_pose, Tn);
        if (it == a.n_iters - 1) {
          float T16[16];
          pose_to_T16(Tn, T16);
          if (SINGLE) {
            PicpState* S = a.state;
            store_pose12(S->pose[0], Tn);
#pragma unroll
            for (int k = 0; k < 16; ++k) S->T16[k] = T16[k];
#pragma unroll
            for (int k = 0; k < 36; ++k) S->H[k] = H[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) S->b[k] = b[k];
            S->chi_in = acc[27]; S->chi_out = acc[28]; S->n_in = (int)acc[29];
          } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) a.T_out[16 * p + k] = T16[k];
            if (a.stats_out) {
              float* so = a.stats_out + 4 * p;
              so[0] = acc[27]; so[1] = acc[28]; so[2] = acc[29]; so[3] = 0.f;
            }
          }
        }
      }
    }
    __syncthreads();
    T = load_pose12(s_pose);
  }
  if (!SINGLE && a.n_iters <= 0 && tid == 0) {          // no round: the starting pose is the result
    float T16[16];
    pose_to_T16(T, T16);
    for (int k = 0; k < 16; ++k) a.T_out[16 * p + k] = T16[k];
    if (a.stats_out) { float* so = a.stats_out + 4 * p; so[0] = so[1] = so[2] = so[3] = 0.f; }
  }
}

hipError_t launch_picp_exact(hipStream_t st, const PicpParams* d_params, PicpState* d_state, PackedCorr pk,
                             int n_iters) {
  if (n_iters <= 0) return hipSuccess;
  ExactArgs a{};
  a.packed = pk.base; a.cap = pk.cap; a.n_pairs = nullptr; a.params = d_params; a.n_iters = n_iters;
  a.state = d_state;
  hipLaunchKernelGGL(picp_exact_kernel<true>, dim3(1), dim3(EX_BLOCK), 0, st, a);
  return hipGetLastError();
}

static hipError_t launch_picp_exact_batch(hipStream_t st, const BatchArgs& b) {
  ExactArgs a{};
  a.packed = b.packed; a.cap = b.cap; a.n_pairs = b.n_pairs; a.params = nullptr;
  a.cam = b.cam; a.thr = b.thr; a.damping = b.damping; a.keep_outliers = b.keep_outliers;
  a.n_iters = b.n_iters; a.state = nullptr; a.T0 = b.T0; a.T_out = b.T_out; a.stats_out = b.stats_out;
  hipLaunchKernelGGL(picp_exact_kernel<false>, dim3(b.n_problems), dim3(EX_BLOCK), 0, st, a);
  return hipGetLastError();
}

// launch-per-round form for a few problems: n_iters + 1 launches, grid (workgroups per problem, problems)
template <bool PINHOLE, bool KEEP>
static void launch_rounds_batch_t(hipStream_t st, const BatchArgs& a) {
  const size_t rows = ((size_t)a.grid + 255) & ~(size_t)255;
  const RoundBatch rb{a.n_pairs, a.cap, PICP_SLOTS * rows * PICP_PSTRIDE * PICP_REPLICAS, a.T_out, a.stats_out};
  const PackedCorr pk{a.packed, a.cap};
  const dim3 g(a.grid, a.n_problems), b(PICP_BLOCK);
  if (a.n_iters <= 0) {       // no round: the starting poses are the result
    hipLaunchKernelGGL(picp_batch_T0_out_kernel, dim3(a.n_problems), dim3(64), 0, st, a);
    return;
  }
  hipLaunchKernelGGL((picp_round_batch_kernel<false, false, PINHOLE, KEEP>), g, b, 0, st, a.params, a.states, pk, a.partials,
                     0, a.grid, rb);
  for (int it = 1; it < a.n_iters; ++it)
    hipLaunchKernelGGL((picp_round_batch_kernel<true, false, PINHOLE, KEEP>), g, b, 0, st, a.params, a.states, pk, a.partials,
                       it, a.grid, rb);
  hipLaunchKernelGGL((picp_round_batch_kernel<true, true, false, false>), dim3(1, a.n_problems), b, 0, st, a.params, a.states,
                     pk, a.partials, a.n_iters, a.grid, rb);
}

// One workgroup per problem (all rounds in one launch) costs ~36 us per round at 50k correspondences whatever the problem
// count (up to one per CU) -- with the other CUs' waves helping (picp_batch_shared_kernel, up to 0.65 problems per CU) 12.6 us
// up to ~60 problems and 0.2 us per problem beyond; the launch-per-round form costs ~5.5 us per round while its workgroups
// fit the chip once and grows with their number.
bool picp_batch_prefers_rounds(int n_problems, size_t cap, int n_iters, int n_cu) {
  if (n_iters <= 0 || n_cu <= 0) return false;
  // per-round cost model fitted on MI355X (tools/batch_forms.py, 50k correspondences: the forms cross at ~10 problems; at ~30
  // before the shared form): launch per round 5.5 us + 1.3 us per further "wave" of workgroups over the CUs
  const double wg_waves = (double)n_problems * (double)picp_grid_for((int)cap, n_cu) / (double)n_cu;
  const double t_rounds = 5.5 + 1.3 * (wg_waves > 1.0 ? wg_waves - 1.0 : 0.0);
  const double size = (double)cap / 50000.0;
  double t_onewg = 36.0 * size * (double)((n_problems + n_cu - 1) / n_cu);
  if (picp_batch_shares(n_problems, cap, n_iters, n_cu)) {
    const double per_problem = 0.2 * (double)n_problems * 256.0 / (double)n_cu;
    t_onewg = 4.0 + size * ((per_problem > 12.6 ? per_problem : 12.6) - 4.0);      // (~4 us of a round are the hand-over, whatever the size)
  }
  return t_rounds < t_onewg;
}

static hipError_t launch_picp_batch_solve(hipStream_t st, const BatchArgs& a);

// The shared form (picp_batch_shared_kernel) for calls that leave CUs without a problem: from 6 trips per problem on (below,
// a home's round is too short for a hand-over to pay).  VO_PICP_SHARE=0 in the environment keeps every call on
// picp_batch_kernel; VO_PICP_HELP_KEEP / _G / _SLACK (tenths of a trip) pin what the kernel otherwise derives.
struct HelpEnv { int share, keep, g, slack10, absent; };
static HelpEnv help_env() {      // read per call (a batched call is milliseconds; tests flip these in-process)
  auto num = [](const char* name, int dflt) { const char* v = getenv(name); return v && v[0] ? atoi(v) : dflt; };
  return HelpEnv{num("VO_PICP_SHARE", 1), num("VO_PICP_HELP_KEEP", 0), num("VO_PICP_HELP_G", 0), num("VO_PICP_HELP_SLACK", 0), num("VO_PICP_HELP_ABSENT", 0)};
}
bool picp_batch_shares(int n_problems, size_t cap, int n_iters, int n_cu) {
  // up to 0.65 problems per CU: beyond, the homes alone already draw what the memory side delivers (200 x 50k: 5.9 TB/s out of
  // the Infinity Cache), and workgroups that add requests without adding bandwidth only add the hand-over (tools/share_ab.py:
  // 160 problems 1.47 against 1.56 ms, 176 problems 1.68 against 1.60)
  return help_env().share != 0 && n_iters > 0 && n_problems >= 1 && n_problems * 100 <= n_cu * 65 && n_problems <= HELP_MAXP &&
         cap >= (size_t)6 * HELP_TRIP && cap < ((size_t)1 << 29);      // (32-bit byte offsets)
}
int picp_help_rows(int n_problems, int n_cu) { return n_cu > n_problems ? (n_cu - n_problems) * HELP_WAVES : 0; }     // one per helper wave
void picp_help_args(BatchArgs& a, unsigned long long* words, int n_cu) {
  a.help_words = words; a.help_grid = words ? n_cu : 0; a.help_rows = words ? picp_help_rows(a.n_problems, n_cu) : 0;
  const HelpEnv e = help_env();
  a.help_keep = e.keep; a.help_g = e.g; a.help_slack10 = e.slack10; a.help_absent = e.absent;
}

hipError_t launch_picp_batch(hipStream_t st, const BatchArgs& a) {
  if (a.n_problems <= 0) return hipSuccess;
  int gx = (int)((a.cap + 255) / 256);
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  BatchArgs ap = a;
  ap.pack_gx = gx;
  if (!a.prepacked) hipLaunchKernelGGL(picp_batch_pack_kernel, frame_grid(gx, a.n_problems), dim3(256), 0, st, ap);
  hipError_t e = launch_picp_batch_solve(st, a);
  if (e == hipSuccess && a.stats_out && a.n_bad) {
    hipLaunchKernelGGL(picp_batch_bad_kernel, dim3((a.n_problems + 255) / 256), dim3(256), 0, st, a);
    e = hipGetLastError();
  }
  return e;
}

static hipError_t launch_picp_batch_solve(hipStream_t st, const BatchArgs& a) {
  if (a.exact) return launch_picp_exact_batch(st, a);
  const bool ph = is_pinhole(a.cam.K), keep = a.keep_outliers != 0;
  if (a.states) {
    if (ph && !keep) launch_rounds_batch_t<true, false>(st, a);
    else if (ph) launch_rounds_batch_t<true, true>(st, a);
    else if (!keep) launch_rounds_batch_t<false, false>(st, a);
    else launch_rounds_batch_t<false, true>(st, a);
    return hipGetLastError();
  }
  if (a.help_words && a.help_grid > a.n_problems) {          // fewer problems than CUs: the others help
    hipError_t e = hipMemsetAsync(a.help_words, 0, sizeof(unsigned long long) * ((size_t)a.help_rows * 32 + (size_t)a.n_problems * 16), st);
    if (e != hipSuccess) return e;
    const dim3 g(a.help_grid), b(PICP_BATCH_BLOCK);
    if (ph && !keep) hipLaunchKernelGGL((picp_batch_shared_kernel<true, false>), g, b, 0, st, a);
    else if (ph) hipLaunchKernelGGL((picp_batch_shared_kernel<true, true>), g, b, 0, st, a);
    else if (!keep) hipLaunchKernelGGL((picp_batch_shared_kernel<false, false>), g, b, 0, st, a);
    else hipLaunchKernelGGL((picp_batch_shared_kernel<false, true>), g, b, 0, st, a);
    return hipGetLastError();
  }
  const dim3 g(a.n_problems), b(PICP_BATCH_BLOCK);
  if (ph && !keep) hipLaunchKernelGGL((picp_batch_kernel<true, false>), g, b, 0, st, a);
  else if (ph) hipLaunchKernelGGL((picp_batch_kernel<true, true>), g, b, 0, st, a);
  else if (!keep) hipLaunchKernelGGL((picp_batch_kernel<false, false>), g, b, 0, st, a);
  else hipLaunchKernelGGL((picp_batch_kernel<false, true>), g, b, 0, st, a);
  return hipGetLastError();
}

}  // namespace vo
