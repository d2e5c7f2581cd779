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
// Exactness: every DECISION (d2 < best, ties) is taken on d2 accumulated exactly
// like the scalar reference loop -- ((t0-q0)^2 + (t1-q1)^2) + ... one term after
// the other, no FMA.  The hot loop in front of it only filters: a fused partial
// sum of the first 3-4 terms against best * (1 + 2^-19), which provably never
// drops a point that could beat or tie the best (see PREFIX_SLACK).  On
// appearance data the filter rejects all but ~5e-4 of the pairs, so the kernel
// issues ~9 instead of ~31 VALU instructions per pair.
//
// Layout: queries live in registers (QPT per thread); tree points are staged
// through LDS in tiles, 12 floats (48 B) per point so that one ds_read_b128
// broadcast fetches the 3-term prefix and two more reads the rest.  The grid
// is (query blocks) x (tree chunks); chunks merge through one 64-bit
// atomicMin per hit on key = (bits(d2) << 32) | tree index, which is
// order-independent, hence deterministic.
#include <stdlib.h>

#include <type_traits>

#include "vo_internal.h"

namespace vo {

hipError_t launch_match_compact(hipStream_t st, const unsigned long long* d_best, int nq, int tree_is_1,
                                int32_t* d_out, int* d_n_out, int* d_scratch, int n_frames, size_t best_stride,
                                size_t out_stride);

constexpr int MB = 256;       // threads per workgroup
constexpr int QPT = 2;        // queries per thread (full scan)
constexpr int QPP = 1;        // queries per thread (pruned scan)
constexpr int MBP = 64;       // threads per workgroup (pruned scan): small query groups keep the scanned rectangle tight
constexpr int TILE = 128;     // tree points per LDS tile (6 KiB)
constexpr int TP = 12;        // padded floats per tree point in LDS
// The hot loop only FILTERS: a fused (FMA) partial sum s' of the first 3-4 squared differences against
// best * PREFIX_SLACK.  With eps = 2^-24 both s' and the reference's unfused partial sum s are within
// (1 +- 4 eps) of the exact real sum, so s' > best * (1 + 2^-19) implies s > best, and the full unfused
// sum (monotone: it only adds non-negative terms to s) is > best as well: a filtered-out point can
// neither beat nor tie the current best.  Every survivor is re-evaluated from scratch in the
// reference's operation order, and only that value takes part in a decision.
constexpr float PREFIX_SLACK = 1.0f + 0x1p-19f;

__global__ __launch_bounds__(256) void match_init_kernel(unsigned long long* best, int nq, float r2, size_t best_stride) {
  best += blockIdx.z * best_stride;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) best[q] = ((unsigned long long)__float_as_uint(r2) << 32) | 0xffffffffull;
}

__global__ __launch_bounds__(MB) void match_kernel(const float* __restrict__ tree, int nt,
                                                   const float* __restrict__ qry, int nq,
                                                   int chunk, float r2,
                                                   unsigned long long* __restrict__ best, size_t tree_stride,
                                                   size_t qry_stride, size_t best_stride) {
  tree += blockIdx.z * tree_stride; qry += blockIdx.z * qry_stride; best += blockIdx.z * best_stride;
  __shared__ __attribute__((aligned(16))) float s_t[TILE * TP];
  const int tid = threadIdx.x;
  const int q0 = (blockIdx.x * MB + tid) * QPT;
  float q[QPT][10];
  float bd[QPT], thr[QPT];
  int bi[QPT];
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const int qi = q0 + j < nq ? q0 + j : (nq > 0 ? nq - 1 : 0);   // clamp: result discarded
    const float2* src = reinterpret_cast<const float2*>(qry + 10 * (size_t)qi);
#pragma unroll
    for (int k = 0; k < 5; ++k) { const float2 v = src[k]; q[j][2 * k] = v.x; q[j][2 * k + 1] = v.y; }
    bd[j] = r2;           // brute_force_search.h:31
    thr[j] = r2 * PREFIX_SLACK;
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
      bool any = false;
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        // conservative filter (fused, 4 terms: ~3e-5 of the pairs survive): see PREFIX_SLACK
        const float d0 = ta.x - q[j][0], d1 = ta.y - q[j][1], d2 = ta.z - q[j][2], d3 = ta.w - q[j][3];
        const float s = __builtin_fmaf(d3, d3, __builtin_fmaf(d2, d2, __builtin_fmaf(d1, d1, d0 * d0)));
        any = any || (s <= thr[j]);
      }
      if (__builtin_expect(any, 0)) {
        const float4 tb4 = *reinterpret_cast<const float4*>(&s_t[p * TP + 4]);
        const float2 tc = *reinterpret_cast<const float2*>(&s_t[p * TP + 8]);
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
          // the decision itself: the reference's unfused left-to-right sum (brute_force_search.h:34)
          float d = ta.x - q[j][0];
          float s = d * d;
          d = ta.y - q[j][1]; s += d * d;
          d = ta.z - q[j][2]; s += d * d;
          d = ta.w - q[j][3]; s += d * d;
          d = tb4.x - q[j][4]; s += d * d;
          d = tb4.y - q[j][5]; s += d * d;
          d = tb4.z - q[j][6]; s += d * d;
          d = tb4.w - q[j][7]; s += d * d;
          d = tc.x - q[j][8]; s += d * d;
          d = tc.y - q[j][9]; s += d * d;
          if (s < bd[j]) { bd[j] = s; thr[j] = s * PREFIX_SLACK; bi[j] = tb + p; }   // brute_force_search.h:35-38
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

// ---- pruned variant ---------------------------------------------------------------
// For large sets the same exact scan runs over a pruned candidate set.  Both
// sets are counting-sorted by a two-level key: NA coarse cells along the
// appearance component A of largest spread, NB fine buckets along the component
// B of second-largest spread (NA*NB = 1024 bins).  A workgroup takes queries of
// ONE cell of A (bucket-sorted in B) and scans, for every A-cell that some query
// could reach, only the B-buckets some query could reach:
//     cells   [min cellA(qA - R), max cellA(qA + R)]
//     buckets [min bktB (qB - R), max bktB (qB + R)]       R = 1.001 * radius
// the minima/maxima taken over the workgroup's queries with the same monotone
// bucket functions the sort used.  A tree point outside that rectangle differs
// from every query of the workgroup by more than the radius in A or in B; that
// single non-negative term of the monotonically accumulated sum already reaches
// radius^2, so the point can never pass the strict "d2 < best" test: pruning
// changes no decision.  Inside the rectangle the scan is the exact one above
// (fused 4-term filter, survivors re-evaluated in the reference's order).
// With no spread in A and B it degenerates to the full scan.
// Batched use: blockIdx.z = frame.  Every per-frame array is base + frame * stride; the single-frame
// entry points launch with gridDim.z = 1 and all strides 0.
struct MatchStrides {
  size_t tree, qry;     // floats between consecutive frames of the two input sets
  size_t ws;            // bytes between consecutive frames' workspaces (identical internal layout)
  size_t best;          // keys between consecutive frames
  size_t mm;            // bytes between consecutive frames' min/max words (kept contiguous: one memset)
};
template <class T>
__device__ __forceinline__ T* frame_ptr(T* p, size_t bytes) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(p)) + bytes);
}

constexpr int NBUCKET = 1024;
constexpr int NA = 16;               // coarse cells
constexpr int NB = NBUCKET / NA;     // fine buckets per cell

struct BucketParams {
  int dimA, dimB;            // appearance components
  float loA, scaleA;         // cell   = clamp((x - loA) * scaleA, 0, NA-1), scaleA = NA / spreadA (0 if none)
  float loB, scaleB;         // bucket = clamp((x - loB) * scaleB, 0, NB-1)
  float R;                   // search margin, slightly above the radius
};

__device__ __forceinline__ int cell_of(float x, float lo, float scale, int n) {
  float v = (x - lo) * scale;                            // monotone non-decreasing in x
  v = fminf(fmaxf(v, 0.f), (float)(n - 1));              // NaN -> 0, +-inf clamp
  return (int)v;
}

__device__ __forceinline__ int bucket_of(float xa, float xb, const BucketParams& bp) {
  return cell_of(xa, bp.loA, bp.scaleA, NA) * NB + cell_of(xb, bp.loB, bp.scaleB, NB);
}

// per-component min/max over both sets: workgroup partials merged with atomic
// min/max on an order-preserving integer image of the floats (NaNs are skipped)
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// mm[0..9] = ord(min), mm[10..19] = ~ord(max): both merge with atomicMin, one 0xff memset initialises all.
// Every sa-th row of a and every sb-th row of b is looked at: the bounds only place the buckets (bucket_of is monotone
// and clamps, so ANY bounds give a correct search), and a sample of a few thousand rows places them as well as all rows.
__global__ __launch_bounds__(256) void match_minmax_kernel(const float* __restrict__ a, int na, int sa,
                                                           const float* __restrict__ b, int nb, int sb, unsigned* mm,
                                                           MatchStrides ms) {
  a += blockIdx.z * ms.tree; b += blockIdx.z * ms.qry; mm = frame_ptr(mm, blockIdx.z * ms.mm);
  __shared__ float s_lo[4][10], s_hi[4][10];
  float lo[10], hi[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
  const int ma = (na + sa - 1) / sa, mb = (nb + sb - 1) / sb;      // sampled rows
  for (int i = blockIdx.x * 256 + threadIdx.x; i < ma + mb; i += gridDim.x * 256) {
    const float2* p = reinterpret_cast<const float2*>(i < ma ? a + 10 * (size_t)i * sa : b + 10 * (size_t)(i - ma) * sb);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float2 v = p[k];
      lo[2 * k] = fminf(lo[2 * k], v.x); hi[2 * k] = fmaxf(hi[2 * k], v.x);          // fmin/fmax drop NaNs
      lo[2 * k + 1] = fminf(lo[2 * k + 1], v.y); hi[2 * k + 1] = fmaxf(hi[2 * k + 1], v.y);
    }
  }
#pragma unroll
  for (int k = 0; k < 10; ++k)
    for (int d = 32; d >= 1; d >>= 1) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], d));
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d));
    }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < 10; ++k) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
  __syncthreads();
  if (threadIdx.x < 10) {
    const int k = threadIdx.x;
    float l = INFINITY, h = -INFINITY;
    for (int w = 0; w < 4; ++w) { l = fminf(l, s_lo[w][k]); h = fmaxf(h, s_hi[w][k]); }
    atomicMin(&mm[k], f2ord(l));
    atomicMin(&mm[10 + k], ~f2ord(h));
  }
}

// Choice of the two bucketing components.  Components 0..3 form the bit-exact early-exit
// prefix of the scan; a component the rectangle has already narrowed makes a poor filter
// term, so A and B are taken from components 4..9 (largest, second-largest finite spread)
// unless those are much flatter (< half the spread) than the best of all ten.
__device__ void top2(const unsigned* mm, int k0, int k1, int& a, int& b, float& sa, float& sb, float& la, float& lb) {
  a = k0; b = k0 + 1; sa = -1.f; sb = -1.f; la = 0.f; lb = 0.f;
  for (int k = k0; k < k1; ++k) {
    const float l = ord2f(mm[k]), h = ord2f(~mm[10 + k]);
    const float span = h - l;
    if (!(span < INFINITY)) continue;                  // empty / infinite / NaN ranges are skipped
    if (span > sa) { sb = sa; b = a; lb = la; sa = span; a = k; la = l; }
    else if (span > sb) { sb = span; b = k; lb = l; }
  }
}

__device__ BucketParams make_bucket_params(const unsigned* __restrict__ mm, float radius) {
  int a, b, a2, b2;
  float sa, sb, la, lb, sa2, sb2, la2, lb2;
  top2(mm, 0, 10, a, b, sa, sb, la, lb);
  top2(mm, 4, 10, a2, b2, sa2, sb2, la2, lb2);
  if (sa2 >= 0.5f * sa && sb2 >= 0.5f * sb) { a = a2; b = b2; sa = sa2; sb = sb2; la = la2; lb = lb2; }
  BucketParams bp;
  bp.dimA = a; bp.loA = la; bp.scaleA = sa > 0.f ? (float)NA / sa : 0.f;
  bp.dimB = b; bp.loB = lb; bp.scaleB = sb > 0.f ? (float)NB / sb : 0.f;
  bp.R = radius * 1.001f;
  return bp;
}

// Counting sort of both sets by bucket without global atomics: SORT_BLOCKS
// workgroups each take a contiguous slice of the combined index space
// [0,nt) tree, [nt,nt+nq) queries.
//   hist    : LDS histogram of the slice -> block_hist[blk][2*NBUCKET]
//   offsets : per bin, exclusive scan over the workgroups and over the bins
//             -> starts[2][NBUCKET+1], block_hist becomes per-(blk,bin) offsets
//   place   : every workgroup ranks its points inside a bin with LDS atomics
//             and writes 48-B records (10 components, original index, bucket)
// The order inside a bucket is arbitrary; nothing downstream depends on it.
constexpr int SORT_BLOCKS = 32;

__device__ __forceinline__ void sort_slice(int nt, int nq, int& lo, int& hi) {
  const int total = nt + nq;
  const int per = (total + SORT_BLOCKS - 1) / SORT_BLOCKS;
  lo = blockIdx.x * per;
  hi = lo + per < total ? lo + per : total;
}

__global__ __launch_bounds__(256) void match_bucket_hist_kernel(const float* __restrict__ tree, int nt,
                                                                const float* __restrict__ qry, int nq,
                                                                const unsigned* __restrict__ mm, float radius,
                                                                BucketParams* bp_out, int* block_hist, MatchStrides ms) {
  tree += blockIdx.z * ms.tree; qry += blockIdx.z * ms.qry;
  mm = frame_ptr(mm, blockIdx.z * ms.mm); bp_out = frame_ptr(bp_out, blockIdx.z * ms.ws);
  block_hist = frame_ptr(block_hist, blockIdx.z * ms.ws);
  __shared__ int s_h[2 * NBUCKET];
  __shared__ BucketParams s_bp;
  for (int k = threadIdx.x; k < 2 * NBUCKET; k += 256) s_h[k] = 0;
  if (threadIdx.x == 0) {                        // every workgroup derives the same parameters from the min/max words
    s_bp = make_bucket_params(mm, radius);
    if (blockIdx.x == 0) *bp_out = s_bp;         // published for the later launches
  }
  __syncthreads();
  const BucketParams bp = s_bp;
  int lo, hi;
  sort_slice(nt, nq, lo, hi);
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const bool is_t = i < nt;
    const float* pt = is_t ? tree + 10 * (size_t)i : qry + 10 * (size_t)(i - nt);
    atomicAdd(&s_h[(is_t ? 0 : NBUCKET) + bucket_of(pt[bp.dimA], pt[bp.dimB], bp)], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * NBUCKET; k += 256) block_hist[(size_t)blockIdx.x * 2 * NBUCKET + k] = s_h[k];
}

// grid 2 (tree half, query half) x NBUCKET threads (one bin each)
__global__ __launch_bounds__(NBUCKET) void match_bucket_offsets_kernel(int* block_hist, int* starts, MatchStrides ms) {
  block_hist = frame_ptr(block_hist, blockIdx.z * ms.ws); starts = frame_ptr(starts, blockIdx.z * ms.ws);
  __shared__ int s_w[NBUCKET / 64];
  const int half = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int h[SORT_BLOCKS];
#pragma unroll
  for (int b = 0; b < SORT_BLOCKS; ++b) h[b] = block_hist[(size_t)b * 2 * NBUCKET + half * NBUCKET + tid];
  int v = 0;
#pragma unroll
  for (int b = 0; b < SORT_BLOCKS; ++b) v += h[b];
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NBUCKET / 64; ++w) { const int c = s_w[w]; if (w < wave) woff += c; tot += c; }
  int run = woff + incl - v;
  starts[half * (NBUCKET + 1) + tid] = run;
  if (tid == 0) starts[half * (NBUCKET + 1) + NBUCKET] = tot;
#pragma unroll
  for (int b = 0; b < SORT_BLOCKS; ++b) {
    block_hist[(size_t)b * 2 * NBUCKET + half * NBUCKET + tid] = run;
    run += h[b];
  }
}

__global__ __launch_bounds__(256) void match_bucket_place_kernel(const float* __restrict__ tree, int nt,
                                                                 const float* __restrict__ qry, int nq,
                                                                 const BucketParams* __restrict__ bpp,
                                                                 const int* __restrict__ block_off, float* tree_rec,
                                                                 float* qry_rec, unsigned long long* best, float r2,
                                                                 MatchStrides ms) {
  tree += blockIdx.z * ms.tree; qry += blockIdx.z * ms.qry; best += blockIdx.z * ms.best;
  bpp = frame_ptr(bpp, blockIdx.z * ms.ws); block_off = frame_ptr(block_off, blockIdx.z * ms.ws);
  tree_rec = frame_ptr(tree_rec, blockIdx.z * ms.ws); qry_rec = frame_ptr(qry_rec, blockIdx.z * ms.ws);
  __shared__ int s_off[2 * NBUCKET];
  for (int k = threadIdx.x; k < 2 * NBUCKET; k += 256) s_off[k] = block_off[(size_t)blockIdx.x * 2 * NBUCKET + k];
  __syncthreads();
  const BucketParams bp = *bpp;
  int lo, hi;
  sort_slice(nt, nq, lo, hi);
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const bool is_t = i < nt;
    const int idx = is_t ? i : i - nt;
    const float2* src = reinterpret_cast<const float2*>((is_t ? tree : qry) + 10 * (size_t)idx);
    float2 v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = src[k];
    float xa = v[0].x, xb = v[0].x;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (bp.dimA == 2 * k) xa = v[k].x;
      if (bp.dimA == 2 * k + 1) xa = v[k].y;
      if (bp.dimB == 2 * k) xb = v[k].x;
      if (bp.dimB == 2 * k + 1) xb = v[k].y;
    }
    const int b = bucket_of(xa, xb, bp);
    const int pos = atomicAdd(&s_off[(is_t ? 0 : NBUCKET) + b], 1);
    float4* dst = reinterpret_cast<float4*>((is_t ? tree_rec : qry_rec) + 12 * (size_t)pos);
    dst[0] = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
    dst[1] = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
    dst[2] = make_float4(v[4].x, v[4].y, __int_as_float(idx), __int_as_float(b));
    if (!is_t) best[idx] = ((unsigned long long)__float_as_uint(r2) << 32) | 0xffffffffull;
  }
}

__global__ __launch_bounds__(MBP) void match_pruned_kernel(const float* __restrict__ tree_rec, int nt,
                                                          const float* __restrict__ qry_rec, int nq,
                                                          const int* __restrict__ starts,
                                                          const BucketParams* __restrict__ bpp, int nchunks, float r2,
                                                          unsigned long long* __restrict__ best, MatchStrides ms) {
  tree_rec = frame_ptr(tree_rec, blockIdx.z * ms.ws); qry_rec = frame_ptr(qry_rec, blockIdx.z * ms.ws);
  starts = frame_ptr(starts, blockIdx.z * ms.ws); bpp = frame_ptr(bpp, blockIdx.z * ms.ws);
  best += blockIdx.z * ms.best;
  __shared__ __attribute__((aligned(16))) float s_t[(TILE + 12) * TP];   // +12: padding of the last trip (<= 7 records) and its look-ahead (4)
  __shared__ int s_rng[4][MBP / 64];
  const int tid = threadIdx.x;
  const int* starts_t = starts;                       // [NBUCKET+1]
  const int* starts_q = starts + (NBUCKET + 1);
  // which A-cell of queries does this workgroup serve?  (workgroups never straddle cells)
  int row = -1, qs = 0, qe = 0;
  {
    int edge[NA + 1];                                  // all cell boundaries at once: independent loads
#pragma unroll
    for (int r = 0; r <= NA; ++r) edge[r] = starts_q[r * NB];
    int wg = blockIdx.x;
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      const int nblk = (edge[r + 1] - edge[r] + MBP * QPP - 1) / (MBP * QPP);
      if (row < 0 && wg < nblk) {
        row = r;
        qs = edge[r] + wg * MBP * QPP;
        qe = qs + MBP * QPP < edge[r + 1] ? qs + MBP * QPP : edge[r + 1];
      }
      if (row < 0) wg -= nblk;
    }
  }
  if (row < 0) return;                                // surplus workgroup (grid is an upper bound)
  const BucketParams bp = *bpp;
  const int q0 = qs + tid * QPP;
  float q[QPP][10];
  float bd[QPP], thr[QPP];
  int bi[QPP], qorig[QPP];
  int aLo = NA, aHi = -1, bLo = NB, bHi = -1;
#pragma unroll
  for (int j = 0; j < QPP; ++j) {
    const bool live = q0 + j < qe;
    const int qi = live ? q0 + j : qe - 1;             // clamp: result discarded
    const float4* src = reinterpret_cast<const float4*>(qry_rec + 12 * (size_t)qi);
    const float4 a = src[0], b = src[1], c = src[2];
    q[j][0] = a.x; q[j][1] = a.y; q[j][2] = a.z; q[j][3] = a.w;
    q[j][4] = b.x; q[j][5] = b.y; q[j][6] = b.z; q[j][7] = b.w;
    q[j][8] = c.x; q[j][9] = c.y;
    qorig[j] = __float_as_int(c.z);
    bd[j] = r2;
    thr[j] = r2 * PREFIX_SLACK;
    bi[j] = -1;
    float xa = q[j][0], xb = q[j][0];
#pragma unroll
    for (int k = 0; k < 10; ++k) { if (bp.dimA == k) xa = q[j][k]; if (bp.dimB == k) xb = q[j][k]; }
    if (live) {
      const int a0 = cell_of(xa - bp.R, bp.loA, bp.scaleA, NA), a1 = cell_of(xa + bp.R, bp.loA, bp.scaleA, NA);
      const int b0 = cell_of(xb - bp.R, bp.loB, bp.scaleB, NB), b1 = cell_of(xb + bp.R, bp.loB, bp.scaleB, NB);
      aLo = a0 < aLo ? a0 : aLo; aHi = a1 > aHi ? a1 : aHi;
      bLo = b0 < bLo ? b0 : bLo; bHi = b1 > bHi ? b1 : bHi;
    }
  }
  // rectangle of the workgroup: min/max over its queries
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int t0 = __shfl_xor(aLo, d), t1 = __shfl_xor(aHi, d), t2 = __shfl_xor(bLo, d), t3 = __shfl_xor(bHi, d);
    aLo = t0 < aLo ? t0 : aLo; aHi = t1 > aHi ? t1 : aHi;
    bLo = t2 < bLo ? t2 : bLo; bHi = t3 > bHi ? t3 : bHi;
  }
  if ((tid & 63) == 0) { s_rng[0][tid >> 6] = aLo; s_rng[1][tid >> 6] = aHi; s_rng[2][tid >> 6] = bLo; s_rng[3][tid >> 6] = bHi; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < MBP / 64; ++w) {
    aLo = s_rng[0][w] < aLo ? s_rng[0][w] : aLo; aHi = s_rng[1][w] > aHi ? s_rng[1][w] : aHi;
    bLo = s_rng[2][w] < bLo ? s_rng[2][w] : bLo; bHi = s_rng[3][w] > bHi ? s_rng[3][w] : bHi;
  }
  // the rectangle is the same in every lane: keep it (and everything derived from it) in scalar
  // registers, so that the tile loop below runs on scalar counters and LDS addresses
  aLo = __builtin_amdgcn_readfirstlane(aLo); aHi = __builtin_amdgcn_readfirstlane(aHi);
  bLo = __builtin_amdgcn_readfirstlane(bLo); bHi = __builtin_amdgcn_readfirstlane(bHi);
  // total candidate count over the cells, split into nchunks slices of whole tiles
  int span = 0;
  for (int r = aLo; r <= aHi; ++r) span += starts_t[r * NB + bHi + 1] - starts_t[r * NB + bLo];
  span = __builtin_amdgcn_readfirstlane(span);
  const int per = ((span + nchunks - 1) / nchunks + TILE - 1) / TILE * TILE;
  const int v_begin = blockIdx.y * per;                // slice of the virtual concatenation of the ranges
  const int v_end = v_begin + per < span ? v_begin + per : span;
  int v0 = 0;                                          // virtual offset of the current range
  for (int r = aLo; r <= aHi; ++r) {
    const int r_begin = __builtin_amdgcn_readfirstlane(starts_t[r * NB + bLo]);
    const int r_len = __builtin_amdgcn_readfirstlane(starts_t[r * NB + bHi + 1]) - r_begin;
    const int lo = v_begin > v0 ? v_begin - v0 : 0;
    const int hi = v_end - v0 < r_len ? v_end - v0 : r_len;
    v0 += r_len;
    for (int off = lo; off < hi; off += TILE) {
      const int tb = r_begin + off;
      const int cnt = hi - off < TILE ? hi - off : TILE;
      __syncthreads();
      const float4* src = reinterpret_cast<const float4*>(tree_rec + 12 * (size_t)tb);
      float4* dst = reinterpret_cast<float4*>(s_t);
      for (int f = tid; f < cnt * 3; f += MBP) dst[f] = src[f];
      // the scan runs 8 points per trip with a look-ahead of 4: the records behind the last point
      // (up to 7 are filtered, 4 more are only loaded) read as "infinitely far", so they never pass
      if (tid < 8) s_t[(cnt + tid) * TP] = __builtin_inff();
      __syncthreads();
      static_assert(QPP == 1, "the grouped scan below keeps one query per lane");
      const float q0x = q[0][0], q1x = q[0][1], q2x = q[0][2], q3x = q[0][3];
      auto filt = [&](const float4& t) {
        // conservative filter (fused, 4 terms): see PREFIX_SLACK
        const float d0 = t.x - q0x, d1 = t.y - q1x, d2 = t.z - q2x, d3 = t.w - q3x;
        return __builtin_fmaf(d3, d3, __builtin_fmaf(d2, d2, __builtin_fmaf(d1, d1, d0 * d0)));
      };
      auto exact = [&](int p) {
        // the decision itself: the reference's unfused left-to-right sum (brute_force_search.h:34)
        const float4 ta = *reinterpret_cast<const float4*>(&s_t[p * TP]);
        const float4 tb4 = *reinterpret_cast<const float4*>(&s_t[p * TP + 4]);
        const float4 tc = *reinterpret_cast<const float4*>(&s_t[p * TP + 8]);
        float d = ta.x - q[0][0];
        float s = d * d;
        d = ta.y - q[0][1]; s += d * d;
        d = ta.z - q[0][2]; s += d * d;
        d = ta.w - q[0][3]; s += d * d;
        d = tb4.x - q[0][4]; s += d * d;
        d = tb4.y - q[0][5]; s += d * d;
        d = tb4.z - q[0][6]; s += d * d;
        d = tb4.w - q[0][7]; s += d * d;
        d = tc.x - q[0][8]; s += d * d;
        d = tc.y - q[0][9]; s += d * d;
        // ties: lowest ORIGINAL index (the sorted order is arbitrary inside a bucket)
        const int ti = __float_as_int(tc.z);
        if (s < bd[0] || (s == bd[0] && bi[0] >= 0 && ti < bi[0])) { bd[0] = s; thr[0] = s * PREFIX_SLACK; bi[0] = ti; }
      };
      auto ld4 = [&](int p, float4& a, float4& b, float4& c, float4& d) {
        const float* r = &s_t[p * TP];
        a = *reinterpret_cast<const float4*>(r); b = *reinterpret_cast<const float4*>(r + TP);
        c = *reinterpret_cast<const float4*>(r + 2 * TP); d = *reinterpret_cast<const float4*>(r + 3 * TP);
      };
      auto scan4 = [&](int p, const float4& t0, const float4& t1, const float4& t2, const float4& t3) {
        const float f0 = filt(t0), f1 = filt(t1), f2 = filt(t2), f3 = filt(t3);
        // <=: an exact tie with a lower original index must still be seen
        const bool pass = (f0 <= thr[0]) | (f1 <= thr[0]) | (f2 <= thr[0]) | (f3 <= thr[0]);
        if (__builtin_expect(__ballot(pass) != 0ull, 0)) {           // wave-uniform: no exec juggling on the fast path
          if (f0 <= thr[0]) exact(p);
          if (f1 <= thr[0]) exact(p + 1);
          if (f2 <= thr[0]) exact(p + 2);
          if (f3 <= thr[0]) exact(p + 3);
        }
      };
      // 8 points per trip in two register sets: while one group of 4 is filtered the next one is in flight
      float4 a0, a1, a2, a3, b0, b1, b2, b3;
      ld4(0, a0, a1, a2, a3);
      for (int p = 0; p < cnt; p += 8) {
        ld4(p + 4, b0, b1, b2, b3);
        scan4(p, a0, a1, a2, a3);
        ld4(p + 8, a0, a1, a2, a3);
        scan4(p + 4, b0, b1, b2, b3);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < QPP; ++j) {
    if (bi[j] >= 0 && q0 + j < qe) {
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(bd[j]) << 32) | (unsigned long long)(unsigned)bi[j];
      atomicMin(&best[qorig[j]], key);
    }
  }
}


static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// per-frame block (sorted records, histograms, bucket starts, parameters)
static size_t match_frame_ws_bytes(int nt, int nq) {
  return align256(sizeof(float) * 12 * ((size_t)nt + (size_t)nq) +
                  sizeof(int) * ((size_t)SORT_BLOCKS * 2 * NBUCKET + 2 * (NBUCKET + 1) + 8) + sizeof(BucketParams) + 512);
}
// whole workspace: the min/max words of all frames first (128 B each), then the per-frame blocks
size_t match_pruned_workspace_bytes(int nt, int nq, int n_frames) {
  return align256(128 * (size_t)n_frames) + (size_t)n_frames * match_frame_ws_bytes(nt, nq);
}

static hipError_t launch_match_pruned(hipStream_t st, const float* tree, int nt, const float* qry, int nq,
                                      float radius, float r2, unsigned long long* d_best, void* ws, int n_cu,
                                      int n_frames, size_t tree_stride, size_t qry_stride, size_t best_stride) {
  // workspace carve of frame 0 (all offsets multiples of 16 bytes); frame f lives ws_stride bytes further
  unsigned* mm = static_cast<unsigned*>(ws);
  char* p = static_cast<char*>(ws) + align256(128 * (size_t)n_frames);
  float* tree_rec = reinterpret_cast<float*>(p); p += sizeof(float) * 12 * (size_t)nt;
  float* qry_rec = reinterpret_cast<float*>(p); p += sizeof(float) * 12 * (size_t)nq;
  int* block_hist = reinterpret_cast<int*>(p); p += sizeof(int) * (size_t)SORT_BLOCKS * 2 * NBUCKET;
  int* starts = reinterpret_cast<int*>(p); p += sizeof(int) * (2 * (NBUCKET + 1) + 6);
  BucketParams* bp = reinterpret_cast<BucketParams*>(p);
  MatchStrides ms;
  ms.tree = tree_stride; ms.qry = qry_stride; ms.best = best_stride;
  ms.ws = n_frames > 1 ? match_frame_ws_bytes(nt, nq) : 0;
  ms.mm = 128;
  const unsigned Z = (unsigned)n_frames;
  hipError_t e = hipMemsetAsync(mm, 0xff, 128 * (size_t)n_frames, st);
  if (e != hipSuccess) return e;
  const int st_t = (nt + 4095) / 4096 > 0 ? (nt + 4095) / 4096 : 1, st_q = (nq + 4095) / 4096 > 0 ? (nq + 4095) / 4096 : 1;   // <= 4096 rows per set
  const int g = ((nt + st_t - 1) / st_t + (nq + st_q - 1) / st_q + 255) / 256;
  hipLaunchKernelGGL(match_minmax_kernel, dim3(g > 0 ? g : 1, 1, Z), dim3(256), 0, st, tree, nt, st_t, qry, nq, st_q, mm, ms);
  hipLaunchKernelGGL(match_bucket_hist_kernel, dim3(SORT_BLOCKS, 1, Z), dim3(256), 0, st, tree, nt, qry, nq, mm, radius,
                     bp, block_hist, ms);
  hipLaunchKernelGGL(match_bucket_offsets_kernel, dim3(2, 1, Z), dim3(NBUCKET), 0, st, block_hist, starts, ms);
  hipLaunchKernelGGL(match_bucket_place_kernel, dim3(SORT_BLOCKS, 1, Z), dim3(256), 0, st, tree, nt, qry, nq, bp,
                     block_hist, tree_rec, qry_rec, d_best, r2, ms);
  const int qblocks = (nq + MBP * QPP - 1) / (MBP * QPP) + NA;   // upper bound: workgroups are aligned to A-cells
  static const int chunk_factor = [] { const char* e = getenv("VO_MATCH_CHUNK_FACTOR"); return e ? atoi(e) : 16; }();
  int nchunks = (chunk_factor * (n_cu > 0 ? n_cu : 256) + qblocks * n_frames - 1) / (qblocks * n_frames);
  if (nchunks < 1) nchunks = 1;
  if (nchunks > 64) nchunks = 64;
  hipLaunchKernelGGL(match_pruned_kernel, dim3(qblocks, nchunks, Z), dim3(MBP), 0, st, tree_rec, nt, qry_rec, nq, starts,
                     bp, nchunks, r2, d_best, ms);
  return hipGetLastError();
}

// ---- cell-hash variant ---------------------------------------------------------------
// Exact search over a 4-D grid.  Four appearance components (largest spreads, taken from components 4..9
// when those are not much flatter, so the filter on components 0..3 keeps its selectivity) are cut into
// nc_k = clamp(floor(spread_k / R), 1, 20) cells of width >= R each (R = 1.001 radius).  The spreads come
// from a strided SAMPLE of both sets (<= CELL_SAMPLE points each): any bounds give a correct grid, because
// the cell function is monotone and clamps -- points beyond the sampled range fall into the edge cells.
// The tree is counting-sorted by the cell key ((c0*nc1 + c1)*nc2 + c2)*nc3 + c3 (<= 160 000 cells) in two
// levels, both on LDS histograms (a global atomic per point -- 64 scattered memory-side requests per wave --
// was 2x dearer): level 1 like the bucket sort above with the <= 400 coarse bins (c0, c1); level 2 one
// workgroup per coarse bin, ordering its points by the <= 400 fine bins (c2, c3) and writing that bin's slice
// of the start table.  The queries are only grouped by coarse bin (level 1 writes their indices).
//
// Search: one workgroup per STRIP of CS_NB coarse bins (c0, c1 .. c1 + CS_NB - 1).  Every query of a bin can only meet
// tree points of the 3 x 3 coarse bins around it, i.e. nine CONTIGUOUS segments of the sorted tree (~140 points each
// on 50k uniform points) and nine rows of the start table: the workgroup stages the 3 x (CS_NB + 2) bins of its strip in
// LDS with coalesced loads (filter prefixes and 16-bit relative starts; the points' original indices stay in global
// memory) and every lane then walks its own <= 27 runs (3 x 3 x 3 cells in c0, c1, c2; contiguous along c3) out of LDS.
// (Per-lane gathers from global memory -- 54 table reads and ~25 16-byte candidate reads per query, each
// its own cache line -- made the first version L1-request-bound: 1.41 ms per 200 x 50k frames.)  A query
// visits, per component, the cells [cell(q - R), cell(q - R) + 2] clipped to cell(q + R): any tree value t
// with |t - q| < radius has cell(q - R) <= cell(t) (monotone) and (t - (q - R)) * scale < (r + R) * scale
// <= 1.9991 (1 + eps) < 2, so cell(t) <= cell(q - R) + 2.  A tree point outside that box differs from the
// query by at least the radius in one component, so that single non-negative term of the monotonically
// accumulated sum already reaches radius^2 and the strict test can never pass: the box changes no decision.
// Inside it the scan is the same conservative filter + exact re-evaluation as above, ties resolved on the
// original index.  On U(-1,1)^10 appearances a query meets ~25 candidates instead of ~1400 in the 2-D
// rectangle.  Segments too long for the LDS budget (clustered data) or a box reaching beyond the staged
// bins (cells narrower than R by a rounding) fall back to the same walk on global memory.
constexpr int HK = 4;                    // hashed components
constexpr int HNC = 20;                  // cells per component, at most
constexpr int HCOARSE = HNC * HNC;       // coarse bins (c0, c1) / fine bins (c2, c3), at most
constexpr int HCPAD = 512;               // HCOARSE rounded up to a power of two (scan width)
constexpr int HBINS = HNC * HNC * HNC * HNC + 1;   // every key + the end sentinel
constexpr int HROW = 408;                // u16 entries per row of the relative start table (>= HCOARSE + 1, 16-byte rows)
constexpr int CELL_MAX_TB = 48, CELL_MAX_QB = 16, CELL_MAX_ROWS = CELL_MAX_TB + CELL_MAX_QB;   // level-1 workgroups per frame, at most
constexpr int CELL_SAMPLE = 2048;        // points per set that define the grid's bounds
#ifndef VO_CS_NB
#define VO_CS_NB 3
#endif
constexpr int CS_NB = VO_CS_NB;          // coarse bins per search workgroup (a strip along c1)
constexpr int CS_THREADS = CS_NB == 1 ? 192 : (CS_NB == 2 ? 320 : 448);   // a coarse bin holds ~140 queries at 50k points
constexpr int CS_SURV = 4;               // filter survivors a lane parks before it evaluates them
constexpr int CS_CAP = CS_NB == 3 ? 2296 : 576 * (CS_NB + 2);   // tree points a search workgroup can stage (16 B each; ~140 per bin
                                                                //   at 50k): with NB = 3 three workgroups share a CU's LDS
static_assert(CS_CAP < 4096 && 3 * (CS_NB + 2) <= 16, "a parked survivor is (staged bin : 4 bits, LDS slot : 12 bits)");

struct CellParams {
  int dim[HK];
  float lo[HK], scale[HK];
  int nc[HK];
  float R;
};

__device__ CellParams make_cell_params(const float* lo_in, const float* hi_in, float radius) {
  float span[10], lo[10];
  for (int k = 0; k < 10; ++k) {
    lo[k] = lo_in[k];
    const float sp = hi_in[k] - lo[k];
    span[k] = (sp < INFINITY) ? sp : -1.f;               // empty / infinite / NaN ranges rank last
  }
  auto top4 = [&](int k0, int* out) {                    // indices of the four largest spans in [k0, 10), descending
    bool used[10] = {false, false, false, false, false, false, false, false, false, false};
    for (int j = 0; j < HK; ++j) {
      int best = -1;
      for (int k = k0; k < 10; ++k)
        if (!used[k] && (best < 0 || span[k] > span[best])) best = k;
      used[best] = true;
      out[j] = best;
    }
  };
  int all4[HK], tail4[HK];
  top4(0, all4);
  top4(4, tail4);
  bool tail_ok = true;
  for (int j = 0; j < HK; ++j) tail_ok = tail_ok && span[tail4[j]] >= 0.5f * span[all4[j]];
  CellParams cp;
  cp.R = radius * 1.001f;
  for (int j = 0; j < HK; ++j) {
    const int k = tail_ok ? tail4[j] : all4[j];
    const float sp = span[k];
    int nc = 1;
    if (sp > 0.f && cp.R > 0.f) {
      const float f = sp / cp.R;
      nc = f >= (float)HNC ? HNC : (int)f;
      if (nc < 1) nc = 1;
    }
    cp.dim[j] = k; cp.lo[j] = lo[k]; cp.nc[j] = nc;
    cp.scale[j] = sp > 0.f ? (float)nc / sp : 0.f;       // cell width sp / nc >= R
  }
  return cp;
}

// component k of v (k is wave-uniform), selected with bit masks: an indexed read -- which is what the compiler makes
// of a chain of selects -- would put v[] into scratch memory
__device__ __forceinline__ float pick10(const float* v, int k) {
  unsigned x = 0;
#pragma unroll
  for (int j = 0; j < 10; ++j) x |= __float_as_uint(v[j]) & (k == j ? 0xffffffffu : 0u);
  return __uint_as_float(x);
}

// per-frame workspace of the cell variant (bytes, every block 256-aligned)
struct CellWs {
  size_t tree_rec, tree_idx, t1_pre, t1_meta, q1_idx, block_hist, coarse_start, start_t, start_rel, cp, total;
};
static CellWs cell_ws_layout(int nt, int nq) {
  CellWs w;
  size_t o = 0;
  w.tree_rec = o; o += align256(sizeof(float) * 4 * (size_t)nt);      // filter prefix (components 0..3) in cell order
  w.tree_idx = o; o += align256(sizeof(int) * (size_t)nt);            // original index of the sorted tree point
  w.t1_pre = o; o += align256(sizeof(float) * 4 * (size_t)nt);        // level 1 (coarse order): tree prefix,
  w.t1_meta = o; o += align256(sizeof(int) * 2 * (size_t)nt);         //   (original index, fine bin)
  w.q1_idx = o; o += align256(sizeof(int) * (size_t)nq);              //   query indices grouped by coarse bin
  w.block_hist = o; o += align256(sizeof(int) * (size_t)CELL_MAX_ROWS * 2 * HCPAD);
  w.coarse_start = o; o += align256(sizeof(int) * 2 * (HCPAD + 1));
  w.start_t = o; o += align256(sizeof(int) * (size_t)HBINS);          // first tree slot of every cell (+ end sentinel)
  w.start_rel = o; o += align256(sizeof(unsigned short) * (size_t)HCOARSE * HROW);   // the same per coarse bin, relative to
                                                                      //   the bin's first slot (16-bit, rows of HROW)
  w.cp = o; o += align256(sizeof(CellParams));
  w.total = o;
  return w;
}
size_t match_cells_workspace_bytes(int nt, int nq, int n_frames) {
  return (size_t)n_frames * cell_ws_layout(nt, nq).total;
}

struct CellArgs {
  const float* tree; const float* qry; int nt, nq;
  char* ws;                 // frame 0's block; frame f = ws + f * ws_stride
  CellWs w;
  size_t ws_stride, tree_stride, qry_stride, best_stride;
  int n_frames;
  int tb, qb;               // level-1 workgroups per frame over the tree / over the queries
  float radius, r2;
  unsigned long long* best;
  int* rs_offsets;          // radius search: [nq + 1] counts, then (after the scan) offsets
  int32_t* rs_indices;      // radius search: tree indices, room for rs_capacity
  int rs_capacity;
};

// XCD-aware decomposition of a 1-D grid of 8 * ceil(n_frames / 8) * per_frame workgroups.  Workgroups are dealt
// round-robin over the 8 XCDs (observed; speed only, nothing depends on it), so giving every frame the workgroups of
// ONE residue class of blockIdx.x mod 8 keeps a frame's working set (4 MB of input, <= 3.6 MB of sorted records) in one
// XCD's L2: partial lines written by the frame's workgroups merge there, segments staged by neighbouring bins hit there.
__device__ __forceinline__ bool xcd_frame_block(int per_frame, int n_frames, int& frame, int& blk) {
  const unsigned L = blockIdx.x, s = L >> 3;
  frame = (int)(s / (unsigned)per_frame) * 8 + (int)(L & 7u);
  blk = (int)(s % (unsigned)per_frame);
  return frame < n_frames;
}
static unsigned xcd_grid(int per_frame, int n_frames) { return 8u * (unsigned)((n_frames + 7) / 8) * (unsigned)per_frame; }

__device__ __forceinline__ void load10(const float* p, float* v) {
  const float2* src = reinterpret_cast<const float2*>(p);
#pragma unroll
  for (int k = 0; k < 5; ++k) { const float2 t = src[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
}
__device__ __forceinline__ void cell_bins(const float* v, const CellParams& cp, int& coarse, int& fine) {
  int c[HK];
#pragma unroll
  for (int j = 0; j < HK; ++j) c[j] = cell_of(pick10(v, cp.dim[j]), cp.lo[j], cp.scale[j], cp.nc[j]);
  coarse = c[0] * cp.nc[1] + c[1];
  fine = c[2] * cp.nc[3] + c[3];
}

// grid bounds: one workgroup per frame, min/max per component over a strided sample of both sets -> CellParams
__global__ __launch_bounds__(1024) void cell_bounds_kernel(CellArgs a) {
  const int f = blockIdx.x;
  const float* tree = a.tree + f * a.tree_stride; const float* qry = a.qry + f * a.qry_stride;
  __shared__ float s_lo[16][10], s_hi[16][10];
  float lo[10], hi[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
  const int st_t = (a.nt + CELL_SAMPLE - 1) / CELL_SAMPLE, st_q = (a.nq + CELL_SAMPLE - 1) / CELL_SAMPLE;
  const int ns_t = st_t ? (a.nt + st_t - 1) / st_t : 0, ns_q = st_q ? (a.nq + st_q - 1) / st_q : 0;
  for (int i = threadIdx.x; i < ns_t + ns_q; i += 1024) {
    const float* row = i < ns_t ? tree + 10 * (size_t)i * st_t : qry + 10 * (size_t)(i - ns_t) * st_q;
    const float2* p = reinterpret_cast<const float2*>(row);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float2 v = p[k];
      lo[2 * k] = fminf(lo[2 * k], v.x); hi[2 * k] = fmaxf(hi[2 * k], v.x);          // fmin/fmax drop NaNs
      lo[2 * k + 1] = fminf(lo[2 * k + 1], v.y); hi[2 * k + 1] = fmaxf(hi[2 * k + 1], v.y);
    }
  }
#pragma unroll
  for (int k = 0; k < 10; ++k)
    for (int d = 32; d >= 1; d >>= 1) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], d));
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d));
    }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < 10; ++k) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l[10], h[10];
    for (int k = 0; k < 10; ++k) {
      l[k] = INFINITY; h[k] = -INFINITY;
      for (int w = 0; w < 16; ++w) { l[k] = fminf(l[k], s_lo[w][k]); h[k] = fmaxf(h[k], s_hi[w][k]); }
    }
    *reinterpret_cast<CellParams*>(a.ws + f * a.ws_stride + a.w.cp) = make_cell_params(l, h, a.radius);
  }
}

// level 1: a.tb workgroups per frame take contiguous slices of the tree, a.qb of the queries (a row of block_hist each;
// only the row's own half is non-zero).  The counts follow the set sizes (cell_level1_blocks) so that a slice fits the
// LDS staging of the placement: 28 + 9 at 50k x 50k; beyond 48 x 1920 tree points the placement writes straight.
constexpr int PL_TCAP = 1920;            // tree points a placement workgroup can order in LDS (24 B each)
constexpr int PL_QCAP = 6400;            // queries (6 B each)
static void cell_level1_blocks(int nt, int nq, int& tb, int& qb) {
  tb = (nt + 1799) / 1800; tb = tb < 4 ? 4 : (tb > CELL_MAX_TB ? CELL_MAX_TB : tb);
  qb = (nq + 5999) / 6000; qb = qb < 2 ? 2 : (qb > CELL_MAX_QB ? CELL_MAX_QB : qb);
}
__device__ __forceinline__ void cell_slice(int blk, int nt, int nq, int tb, int qb, bool& is_t, int& lo, int& hi) {
  is_t = blk < tb;
  const int n = is_t ? nt : nq, parts = is_t ? tb : qb, b = is_t ? blk : blk - tb;
  const int per = (n + parts - 1) / parts;
  lo = b * per < n ? b * per : n;
  hi = lo + per < n ? lo + per : n;
}

__global__ __launch_bounds__(256) void cell_coarse_hist_kernel(CellArgs a) {
  int f, blk;
  if (!xcd_frame_block(a.tb + a.qb, a.n_frames, f, blk)) return;
  char* ws = a.ws + f * a.ws_stride;
  __shared__ int s_h[HCPAD];
  for (int k = threadIdx.x; k < HCPAD; k += 256) s_h[k] = 0;
  const CellParams cp = *reinterpret_cast<const CellParams*>(ws + a.w.cp);
  __syncthreads();
  bool is_t; int lo, hi;
  cell_slice(blk, a.nt, a.nq, a.tb, a.qb, is_t, lo, hi);
  const float* src = is_t ? a.tree + f * a.tree_stride : a.qry + f * a.qry_stride;
  // only the two coarse components are needed here (wave-uniform column indices): two 4-byte loads per point instead of the row
  const int d0 = cp.dim[0], d1 = cp.dim[1];
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const float x0 = src[10 * (size_t)i + d0], x1 = src[10 * (size_t)i + d1];
    const int coarse = cell_of(x0, cp.lo[0], cp.scale[0], cp.nc[0]) * cp.nc[1] + cell_of(x1, cp.lo[1], cp.scale[1], cp.nc[1]);
    atomicAdd(&s_h[coarse], 1);
  }
  __syncthreads();
  int* row = reinterpret_cast<int*>(ws + a.w.block_hist) + (size_t)blk * 2 * HCPAD;
  for (int k = threadIdx.x; k < HCPAD; k += 256) { row[(is_t ? 0 : HCPAD) + k] = s_h[k]; row[(is_t ? HCPAD : 0) + k] = 0; }
}

// level 1, offsets: grid 2 (tree half, query half) x HCPAD threads (one coarse bin each): exclusive scan over the
// workgroups and over the bins -> coarse_start[2][HCPAD+1]; block_hist becomes per-(workgroup, bin) offsets
__global__ __launch_bounds__(HCPAD) void cell_coarse_offsets_kernel(CellArgs a) {
  const int f = blockIdx.z;
  char* ws = a.ws + f * a.ws_stride;
  int* block_hist = reinterpret_cast<int*>(ws + a.w.block_hist);
  int* starts = reinterpret_cast<int*>(ws + a.w.coarse_start);
  __shared__ int s_w[HCPAD / 64];
  const int half = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b_lo = half ? a.tb : 0, b_hi = half ? a.tb + a.qb : a.tb;          // the rows that hold this set
  int v = 0;
  for (int b = b_lo; b < b_hi; ++b) v += block_hist[(size_t)b * 2 * HCPAD + half * HCPAD + tid];
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < HCPAD / 64; ++w) { const int c = s_w[w]; if (w < wave) woff += c; tot += c; }
  int run = woff + incl - v;
  starts[half * (HCPAD + 1) + tid] = run;
  if (tid == 0) starts[half * (HCPAD + 1) + HCPAD] = tot;
  for (int b = b_lo; b < b_hi; ++b) {
    int* slot = &block_hist[(size_t)b * 2 * HCPAD + half * HCPAD + tid];
    const int h = *slot;
    *slot = run;
    run += h;
  }
}

// level 1, placement in coarse order.  Tree: filter prefix + (original index, fine bin); queries: the index alone.
// A workgroup's records land in ~400 bins, four or so per bin: written straight from the registers that is one
// scattered 16/8/4-byte store per lane, and the L2 channels take those one request at a time (0.28 of the kernel's
// 0.43 ms per 200 frames).  The workgroup therefore ORDERS its records by bin in LDS first (its own counts per bin are
// the differences of consecutive rows of offsets) and then copies LDS -> global in that order: consecutive lanes write
// consecutive addresses inside a bin's run, and the frame's workgroups share one XCD, whose L2 merges the runs.
__global__ __launch_bounds__(256) void cell_coarse_place_kernel(CellArgs a) {
  int f, blk;
  if (!xcd_frame_block(a.tb + a.qb, a.n_frames, f, blk)) return;
  char* ws = a.ws + f * a.ws_stride;
  bool is_t; int lo, hi;
  cell_slice(blk, a.nt, a.nq, a.tb, a.qb, is_t, lo, hi);
  const float* src = is_t ? a.tree + f * a.tree_stride : a.qry + f * a.qry_stride;
  const int half = is_t ? 0 : HCPAD, set = is_t ? 0 : 1;
  const int* block_off = reinterpret_cast<const int*>(ws + a.w.block_hist);
  const int* cstart = reinterpret_cast<const int*>(ws + a.w.coarse_start) + set * (HCPAD + 1);
  float4* t1_pre = reinterpret_cast<float4*>(ws + a.w.t1_pre);
  int2* t1_meta = reinterpret_cast<int2*>(ws + a.w.t1_meta);
  int* q1_idx = reinterpret_cast<int*>(ws + a.w.q1_idx);
  __shared__ int s_off[HCPAD];           // first global slot of this workgroup's records, per bin
  __shared__ int s_cur[HCPAD];           // LDS cursor per bin (starts at the bin's first LDS slot)
  __shared__ int s_delta[HCPAD];         // global slot - LDS slot, per bin
  __shared__ int s_w[4];
  __shared__ __attribute__((aligned(16))) unsigned char s_rec[PL_TCAP * 24];   // tree: float4 prefix[PL_TCAP] + int2 meta[PL_TCAP];
                                                                               // queries: int idx[PL_QCAP] + ushort bin[PL_QCAP]
  static_assert(PL_QCAP * 6 <= PL_TCAP * 24, "query staging must fit the tree staging");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool last_row = is_t ? blk == a.tb - 1 : blk == a.tb + a.qb - 1;
  // this workgroup's count per bin = next row's offset (or the bin's end) - its own offset; two bins per thread
  int off[2], cnt[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = 2 * tid + j;
    off[j] = block_off[(size_t)blk * 2 * HCPAD + half + k];
    const int nxt = last_row ? (k + 1 <= HCPAD ? cstart[k + 1] : off[j]) : block_off[(size_t)(blk + 1) * 2 * HCPAD + half + k];
    cnt[j] = nxt - off[j];
  }
  const int v = cnt[0] + cnt[1];
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { const int c = s_w[w]; if (w < wave) woff += c; total += c; }
  const int ex = woff + incl - v;
  s_off[2 * tid] = off[0]; s_off[2 * tid + 1] = off[1];
  s_cur[2 * tid] = ex; s_cur[2 * tid + 1] = ex + cnt[0];
  s_delta[2 * tid] = off[0] - ex; s_delta[2 * tid + 1] = off[1] - (ex + cnt[0]);
  const CellParams cp = *reinterpret_cast<const CellParams*>(ws + a.w.cp);
  const bool ordered = total == hi - lo && total <= (is_t ? PL_TCAP : PL_QCAP);     // else: straight stores
  __syncthreads();
  float4* r_pre = reinterpret_cast<float4*>(s_rec);
  int2* r_meta = reinterpret_cast<int2*>(s_rec + (size_t)PL_TCAP * 16);
  int* r_qidx = reinterpret_cast<int*>(s_rec);
  unsigned short* r_qbin = reinterpret_cast<unsigned short*>(s_rec + (size_t)PL_QCAP * 4);
  for (int i = lo + tid; i < hi; i += 256) {
    float v10[10];
    load10(src + 10 * (size_t)i, v10);
    int coarse, fine;
    cell_bins(v10, cp, coarse, fine);
    if (ordered) {
      const int slot = atomicAdd(&s_cur[coarse], 1);
      if (is_t) {
        r_pre[slot] = make_float4(v10[0], v10[1], v10[2], v10[3]);
        r_meta[slot] = make_int2(i, fine | (coarse << 16));
      } else {
        r_qidx[slot] = i;
        r_qbin[slot] = (unsigned short)coarse;
      }
    } else {
      const int pos = atomicAdd(&s_off[coarse], 1);
      if (is_t) { t1_pre[pos] = make_float4(v10[0], v10[1], v10[2], v10[3]); t1_meta[pos] = make_int2(i, fine); }
      else q1_idx[pos] = i;
    }
  }
  if (!ordered) return;
  __syncthreads();
  for (int i = tid; i < total; i += 256) {
    if (is_t) {
      const int2 m = r_meta[i];
      const int dest = i + s_delta[m.y >> 16];
      t1_pre[dest] = r_pre[i];
      t1_meta[dest] = make_int2(m.x, m.y & 0xffff);
    } else {
      q1_idx[i + s_delta[r_qbin[i]]] = r_qidx[i];
    }
  }
}

// level 2 (tree): one workgroup per coarse bin: counting sort of the bin's points by fine bin in LDS, and the bin's
// slice of the start table (absolute slots) -- the last coarse bin adds the sentinel
__global__ __launch_bounds__(256) void cell_fine_kernel(CellArgs a) {
  int f, coarse;
  if (!xcd_frame_block(HCOARSE, a.n_frames, f, coarse)) return;
  char* ws = a.ws + f * a.ws_stride;
  const CellParams cp = *reinterpret_cast<const CellParams*>(ws + a.w.cp);
  const int n_coarse = cp.nc[0] * cp.nc[1], n_fine = cp.nc[2] * cp.nc[3];
  if (coarse >= n_coarse) return;
  const int* cstart = reinterpret_cast<const int*>(ws + a.w.coarse_start);
  const int begin = cstart[coarse], end = cstart[coarse + 1];
  __shared__ int s_cnt[HCPAD];
  __shared__ int s_w[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < HCPAD; k += 256) s_cnt[k] = 0;
  __syncthreads();
  const int2* t1_meta = reinterpret_cast<const int2*>(ws + a.w.t1_meta);
  for (int i = begin + tid; i < end; i += 256) atomicAdd(&s_cnt[t1_meta[i].y], 1);
  __syncthreads();
  // exclusive scan of the HCPAD counters: two per thread
  const int c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
  const int v = c0 + c1;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int woff = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) if (w < wave) woff += s_w[w];
  const int ex = begin + woff + incl - v;
  __syncthreads();
  s_cnt[2 * tid] = ex; s_cnt[2 * tid + 1] = ex + c0;               // cursors = absolute first slots
  int* start_t = reinterpret_cast<int*>(ws + a.w.start_t) + (size_t)coarse * n_fine;
  if (2 * tid < n_fine) start_t[2 * tid] = ex;
  if (2 * tid + 1 < n_fine) start_t[2 * tid + 1] = ex + c0;
  if (coarse == n_coarse - 1 && tid == 0) start_t[n_fine] = end;   // end sentinel (= nt)
  // the same row relative to the bin's first slot, 16 bits per entry (what the search stages; a bin of >= 65536
  // points saturates and is searched through the absolute table instead)
  unsigned short* rel = reinterpret_cast<unsigned short*>(ws + a.w.start_rel) + (size_t)coarse * HROW;
  {
    const int r0 = ex - begin, r1 = ex + c0 - begin;
    if (2 * tid < n_fine) rel[2 * tid] = (unsigned short)(r0 < 65535 ? r0 : 65535);
    if (2 * tid + 1 < n_fine) rel[2 * tid + 1] = (unsigned short)(r1 < 65535 ? r1 : 65535);
    if (tid == 0) { const int re = end - begin; rel[n_fine] = (unsigned short)(re < 65535 ? re : 65535); }
  }
  __syncthreads();
  const float4* t1_pre = reinterpret_cast<const float4*>(ws + a.w.t1_pre);
  float4* tree_pre = reinterpret_cast<float4*>(ws + a.w.tree_rec);
  int* tree_idx = reinterpret_cast<int*>(ws + a.w.tree_idx);
  for (int i = begin + tid; i < end; i += 256) {
    const int2 m = t1_meta[i];
    const int pos = atomicAdd(&s_cnt[m.y], 1);
    tree_pre[pos] = t1_pre[i];
    tree_idx[pos] = m.x;
  }
}

// MODE 0: best match per query (bestMatchFull);  MODE 1 / 2: count / write ALL tree points with d2 < r2
// (fullSearch, eigen_kdtree.h:56-71 + bruteForceSearch, brute_force_search.h:3-20)
// One workgroup serves a STRIP of CS_NB coarse bins (c0, c1f .. c1f + CS_NB - 1): their queries are contiguous in
// q1_idx, and their neighbourhoods overlap -- 3 x (CS_NB + 2) staged bins instead of 9 per bin.
constexpr int CS_PLANES = 3 * (CS_NB + 2);
constexpr int CS_STRIPS = (HNC + CS_NB - 1) / CS_NB;     // strips per c0 row, at most
template <int MODE>
__global__ __launch_bounds__(CS_THREADS) void cell_search_kernel(CellArgs a) {
  int f, blk;
  if (!xcd_frame_block(HNC * CS_STRIPS, a.n_frames, f, blk)) return;
  char* ws = a.ws + f * a.ws_stride;
  const CellParams cp = *reinterpret_cast<const CellParams*>(ws + a.w.cp);
  const int n0 = cp.nc[0], n1 = cp.nc[1], n2 = cp.nc[2], n3 = cp.nc[3];
  const int n_fine = n2 * n3;
  const int c0 = blk / CS_STRIPS, c1f = (blk - c0 * CS_STRIPS) * CS_NB;
  if (c0 >= n0 || c1f >= n1) return;
  const int nb_here = n1 - c1f < CS_NB ? n1 - c1f : CS_NB;
  const int coarse0 = c0 * n1 + c1f;
  const int* __restrict__ cstart_t = reinterpret_cast<const int*>(ws + a.w.coarse_start);
  const int* __restrict__ cstart_q = cstart_t + (HCPAD + 1);
  const int qb = cstart_q[coarse0], qe = cstart_q[coarse0 + nb_here];
  if (qb >= qe) return;                                   // no query lives here: nothing to stage
  const int* __restrict__ start_t = reinterpret_cast<const int*>(ws + a.w.start_t);
  const float4* __restrict__ tree_pre = reinterpret_cast<const float4*>(ws + a.w.tree_rec);
  const int* __restrict__ tree_idx = reinterpret_cast<const int*>(ws + a.w.tree_idx);
  const int* __restrict__ q1_idx = reinterpret_cast<const int*>(ws + a.w.q1_idx);
  const float* __restrict__ tree = a.tree + f * a.tree_stride;
  const float* __restrict__ qry = a.qry + f * a.qry_stride;
  unsigned long long* best = a.best + f * a.best_stride;
  const int tid = threadIdx.x;

  __shared__ float4 s_pre[CS_CAP];                        // filter prefixes of the staged tree points (their indices stay in
                                                          //   global memory: only the ~3 survivors per query need one)
  __shared__ __attribute__((aligned(16))) unsigned short s_start[CS_PLANES][HROW];   // cell starts of the staged bins, relative to each bin's first slot
  __shared__ int s_gbase[CS_PLANES], s_lbase[CS_PLANES + 1], s_len[CS_PLANES];   // per bin: first slot in the sorted tree, in LDS, length
  __shared__ unsigned short s_surv[CS_SURV][CS_THREADS];  // parked filter survivors (LDS slots), lane-private columns
  // staged bin `s` = (row, col): coarse bin (c0 - 1 + row, c1f - 1 + col); bins outside the grid have length 0
  if (tid < CS_PLANES) {
    const int p0 = c0 - 1 + tid / (CS_NB + 2), p1 = c1f - 1 + tid % (CS_NB + 2);
    int gb = 0, len = 0;
    if (p0 >= 0 && p0 < n0 && p1 >= 0 && p1 < n1) { gb = cstart_t[p0 * n1 + p1]; len = cstart_t[p0 * n1 + p1 + 1] - gb; }
    s_gbase[tid] = gb; s_len[tid] = len;
  }
  // the first batch of queries: their (dependent) loads are in flight during the set-up and the staging
  int qi = qb + tid;
  int qorig = qi < qe ? q1_idx[qi] : q1_idx[qb];
  float q[10];
  load10(qry + 10 * (size_t)qorig, q);
  __syncthreads();
  int total = 0, longest = 0;
  if (tid == 0) s_lbase[0] = 0;
#pragma unroll
  for (int s = 0; s < CS_PLANES; ++s) {
    const int len = s_len[s];
    total += len;
    longest = len > longest ? len : longest;
    if (tid == 0) s_lbase[s + 1] = total;
  }
  const bool staged = total <= CS_CAP && longest < 65535;
  __syncthreads();
  if (staged) {
    // flattened over the staged bins: every thread's loads are independent, several in flight at once
#pragma unroll 4
    for (int i = tid; i < total; i += CS_THREADS) {
      int s = 0;
#pragma unroll
      for (int j = 1; j < CS_PLANES; ++j) s += (i >= s_lbase[j]) ? 1 : 0;   // empty bins share a base: the last one wins,
      const int g = s_gbase[s] + (i - s_lbase[s]);                          // and only a bin with points can own slot i
      s_pre[i] = tree_pre[g];
    }
    // the bins' rows of the 16-bit start table, eight entries per load
    const uint4* rel = reinterpret_cast<const uint4*>(ws + a.w.start_rel);
    const int vec_per_row = (n_fine + 1 + 7) / 8;
#pragma unroll 2
    for (int i = tid; i < CS_PLANES * (HROW / 8); i += CS_THREADS) {
      const int s = i / (HROW / 8), v = i - s * (HROW / 8);
      if (s_len[s] > 0 && v < vec_per_row) {
        const int cb = (c0 - 1 + s / (CS_NB + 2)) * n1 + (c1f - 1 + s % (CS_NB + 2));
        reinterpret_cast<uint4*>(&s_start[s][0])[v] = rel[(size_t)cb * (HROW / 8) + v];
      }
    }
  }
  __syncthreads();

  for (; qi < qe; qi += CS_THREADS) {
    if (qi >= qb + CS_THREADS) { qorig = q1_idx[qi]; load10(qry + 10 * (size_t)qorig, q); }   // later batches (rare)
    int c_lo[HK], c_hi[HK];
#pragma unroll
    for (int j = 0; j < HK; ++j) {
      const float x = pick10(q, cp.dim[j]);
      c_lo[j] = cell_of(x - cp.R, cp.lo[j], cp.scale[j], cp.nc[j]);
      const int h = cell_of(x + cp.R, cp.lo[j], cp.scale[j], cp.nc[j]);
      c_hi[j] = h < c_lo[j] + 2 ? h : c_lo[j] + 2;
    }
    float bd = a.r2, thr = a.r2 * PREFIX_SLACK;
    int bi = -1;
    int n_hit = 0;
    int out_at = 0;
    if (MODE == 2) out_at = a.rs_offsets[qorig];
    // the decision itself: the reference's unfused left-to-right sum (brute_force_search.h:34) over the whole row
    auto decide = [&](const float4 ta, const int ti, const float2 r2v, const float2 r3v, const float2 r4v) {
      float d = ta.x - q[0];
      float s = d * d;
      d = ta.y - q[1]; s += d * d;
      d = ta.z - q[2]; s += d * d;
      d = ta.w - q[3]; s += d * d;
      d = r2v.x - q[4]; s += d * d;
      d = r2v.y - q[5]; s += d * d;
      d = r3v.x - q[6]; s += d * d;
      d = r3v.y - q[7]; s += d * d;
      d = r4v.x - q[8]; s += d * d;
      d = r4v.y - q[9]; s += d * d;
      if (MODE == 0) {
        if (s < bd || (s == bd && bi >= 0 && ti < bi)) { bd = s; thr = s * PREFIX_SLACK; bi = ti; }
      } else if (s < bd) {                               // bd stays radius^2: every point inside the ball
        if (MODE == 2 && out_at + n_hit < a.rs_capacity) a.rs_indices[out_at + n_hit] = ti;
        ++n_hit;
      }
    };
    // conservative filter (fused, 4 terms): see PREFIX_SLACK.  <=: an exact tie with a lower original index must be seen
    auto passes = [&](const float4 ta) {
      const float d0 = ta.x - q[0], d1 = ta.y - q[1], d2 = ta.z - q[2], d3 = ta.w - q[3];
      return __builtin_fmaf(d3, d3, __builtin_fmaf(d2, d2, __builtin_fmaf(d1, d1, d0 * d0))) <= thr;
    };
    auto consider = [&](const float4 ta, const int ti) {
      if (passes(ta)) {
        const float2* row = reinterpret_cast<const float2*>(tree + 10 * (size_t)ti);
        decide(ta, ti, row[2], row[3], row[4]);
      }
    };
    // the query's box in staged-bin coordinates
    const int row_lo = c_lo[0] - (c0 - 1), row_hi = c_hi[0] - (c0 - 1);
    const int col_lo = c_lo[1] - (c1f - 1), col_hi = c_hi[1] - (c1f - 1);
    if (staged && row_lo >= 0 && row_hi <= 2 && col_lo >= 0 && col_hi <= CS_NB + 1) {
      // Walk first, decide afterwards.  A survivor of the filter needs the rest of its row from global memory (a
      // miss of ~1 us); deciding it on the spot would stall the whole wave once per lane.  The walk therefore only
      // PARKS survivors (their LDS slot); the rows of all lanes' k-th survivors are then fetched together.  The
      // filter runs against radius^2 throughout (it cannot tighten before a decision): a few more survivors
      // (~2.5 per query on uniform data), no different result -- every decision is order-independent
      // (minimum of (d2, index)).  A lane whose list is full decides that survivor on the spot.
      int n_surv = 0;
      // <= 3 x 3 staged bins per lane; per bin the lane's three runs (c2 = lo..lo+2, contiguous along c3) are walked
      // as ONE flattened sequence: the trip count of the wave is the largest sum of three run lengths among its
      // lanes, not the sum of three maxima.  The next candidate is fetched while the current one is filtered.
      for (int i0 = 0; i0 < 3; ++i0)
        for (int i1 = 0; i1 < 3; ++i1) {
          const bool in01 = row_lo + i0 <= row_hi && col_lo + i1 <= col_hi;
          const int s = in01 ? (row_lo + i0) * (CS_NB + 2) + col_lo + i1 : 0;
          // the three runs as plain scalars, and no lambda that captures them by reference: the compiler otherwise
          // parks them in scratch memory and selects among their ADDRESSES
          const bool in_s = in01 && s_len[s] > 0;
          const int lb = s_lbase[s];
          const int c2a = c_lo[2], c2b = c_lo[2] + 1, c2c = c_lo[2] + 2;
          const bool ina = in_s && c2a <= c_hi[2], inb = in_s && c2b <= c_hi[2], inc = in_s && c2c <= c_hi[2];
          const int sa0 = s_start[s][ina ? c2a * n3 + c_lo[3] : 0], sb0 = s_start[s][ina ? c2a * n3 + c_hi[3] + 1 : 0];
          const int sa1 = s_start[s][inb ? c2b * n3 + c_lo[3] : 0], sb1 = s_start[s][inb ? c2b * n3 + c_hi[3] + 1 : 0];
          const int sa2 = s_start[s][inc ? c2c * n3 + c_lo[3] : 0], sb2 = s_start[s][inc ? c2c * n3 + c_hi[3] + 1 : 0];
          const int ln0 = ina ? sb0 - sa0 : 0, ln1 = inb ? sb1 - sa1 : 0, ln2 = inc ? sb2 - sa2 : 0;
          const int l01 = ln0 + ln1, tot = l01 + ln2;
          const int b0 = lb + sa0, b1 = lb + sa1 - ln0, b2 = lb + sa2 - l01;     // slot(k) = k + (b0 | b1 | b2)
          int pos = b0;
          pos = 0 >= ln0 ? b1 : pos;
          pos = 0 >= l01 ? b2 : pos;
          pos = tot > 0 ? pos : 0;
          float4 ta = s_pre[pos];
          for (int k = 0; k < tot; ++k) {
            int base_n = b0;
            base_n = k + 1 >= ln0 ? b1 : base_n;
            base_n = k + 1 >= l01 ? b2 : base_n;
            const int pos_n = k + 1 < tot ? base_n + k + 1 : pos;       // (always a valid slot: one unconditional read)
            const float4 ta_n = s_pre[pos_n];
            if (passes(ta)) {
              if (n_surv < CS_SURV) { s_surv[n_surv][tid] = (unsigned short)(pos | (s << 12)); ++n_surv; }
              else consider(ta, tree_idx[s_gbase[s] + pos - lb]);
            }
            pos = pos_n; ta = ta_n;
          }
        }
      for (int k = 0; k < n_surv; k += 3) {               // three rows per lane in flight
        const bool h1 = k + 1 < n_surv, h2 = k + 2 < n_surv;
        const int ka = s_surv[k][tid], kb = s_surv[h1 ? k + 1 : k][tid], kc = s_surv[h2 ? k + 2 : k][tid];
        const int pa = ka & 4095, pb = kb & 4095, pc = kc & 4095;            // LDS slots; bits 15:12 = the staged bin
        const float4 fa = s_pre[pa], fb = s_pre[pb], fc = s_pre[pc];
        const int ia = tree_idx[s_gbase[ka >> 12] + pa - s_lbase[ka >> 12]];
        const int ib = tree_idx[s_gbase[kb >> 12] + pb - s_lbase[kb >> 12]];
        const int ic = tree_idx[s_gbase[kc >> 12] + pc - s_lbase[kc >> 12]];
        const float2* ra = reinterpret_cast<const float2*>(tree + 10 * (size_t)ia);
        const float2* rb = reinterpret_cast<const float2*>(tree + 10 * (size_t)ib);
        const float2* rc = reinterpret_cast<const float2*>(tree + 10 * (size_t)ic);
        const float2 a2 = ra[2], a3 = ra[3], a4 = ra[4], b2 = rb[2], b3 = rb[3], b4 = rb[4], c2 = rc[2], c3 = rc[3], c4 = rc[4];
        decide(fa, ia, a2, a3, a4);
        if (h1) decide(fb, ib, b2, b3, b4);
        if (h2) decide(fc, ic, c2, c3, c4);
      }
    } else {
      // the same walk on global memory (segments beyond the LDS budget, or a box beyond the staged bins)
      for (int x0 = c_lo[0]; x0 <= c_hi[0]; ++x0)
        for (int x1 = c_lo[1]; x1 <= c_hi[1]; ++x1)
          for (int x2 = c_lo[2]; x2 <= c_hi[2]; ++x2) {
            const int key0 = ((x0 * n1 + x1) * n2 + x2) * n3;
            const int e = start_t[key0 + c_hi[3] + 1];
            for (int p = start_t[key0 + c_lo[3]]; p < e; ++p) consider(tree_pre[p], tree_idx[p]);
          }
    }
    if (MODE == 0)
      best[qorig] = bi >= 0 ? (((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)(unsigned)bi)
                            : (((unsigned long long)__float_as_uint(a.r2) << 32) | 0xffffffffull);
    if (MODE == 1) a.rs_offsets[qorig] = n_hit;
  }
}

static hipError_t launch_cells_sort(hipStream_t st, CellArgs& a, const float* tree, int nt, const float* qry, int nq,
                                    float radius, float r2, unsigned long long* d_best, void* ws, int n_frames,
                                    size_t tree_stride, size_t qry_stride, size_t best_stride) {
  a.tree = tree; a.qry = qry; a.nt = nt; a.nq = nq;
  a.ws = static_cast<char*>(ws);
  a.w = cell_ws_layout(nt, nq);
  a.ws_stride = a.w.total; a.tree_stride = tree_stride; a.qry_stride = qry_stride; a.best_stride = best_stride;
  a.n_frames = n_frames;
  cell_level1_blocks(nt, nq, a.tb, a.qb);
  a.radius = radius; a.r2 = r2; a.best = d_best;
  a.rs_offsets = nullptr; a.rs_indices = nullptr; a.rs_capacity = 0;
  const unsigned Z = (unsigned)n_frames;
  hipLaunchKernelGGL(cell_bounds_kernel, dim3(Z), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(cell_coarse_hist_kernel, dim3(xcd_grid(a.tb + a.qb, n_frames)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(cell_coarse_offsets_kernel, dim3(2, 1, Z), dim3(HCPAD), 0, st, a);
  hipLaunchKernelGGL(cell_coarse_place_kernel, dim3(xcd_grid(a.tb + a.qb, n_frames)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(cell_fine_kernel, dim3(xcd_grid(HCOARSE, n_frames)), dim3(256), 0, st, a);
  return hipGetLastError();
}

static hipError_t launch_match_cells(hipStream_t st, const float* tree, int nt, const float* qry, int nq,
                                     float radius, float r2, unsigned long long* d_best, void* ws, int n_frames,
                                     size_t tree_stride, size_t qry_stride, size_t best_stride) {
  CellArgs a;
  hipError_t e = launch_cells_sort(st, a, tree, nt, qry, nq, radius, r2, d_best, ws, n_frames, tree_stride, qry_stride,
                                   best_stride);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cell_search_kernel<0>, dim3(xcd_grid(HNC * CS_STRIPS, n_frames)), dim3(CS_THREADS), 0, st, a);
  return hipGetLastError();
}

// fullSearch for every query: d_offsets[nq + 1] (CSR), d_indices[capacity] (tree indices, order unspecified inside a
// query's list); d_offsets[nq] = number of hits found, also when it exceeds the capacity (the surplus is dropped)
hipError_t launch_radius_search(hipStream_t st, const float* d_tree, int nt, const float* d_qry, int nq, float radius,
                                int* d_offsets, int32_t* d_indices, int capacity, void* ws) {
  if (nq <= 0 || nt <= 0) return hipMemsetAsync(d_offsets, 0, sizeof(int) * (size_t)((nq > 0 ? nq : 0) + 1), st);
  CellArgs a;
  hipError_t e = launch_cells_sort(st, a, d_tree, nt, d_qry, nq, radius, radius * radius, nullptr, ws, 1, 0, 0, 0);
  if (e != hipSuccess) return e;
  a.rs_offsets = d_offsets; a.rs_indices = d_indices; a.rs_capacity = capacity;
  hipLaunchKernelGGL(cell_search_kernel<1>, dim3(xcd_grid(HNC * CS_STRIPS, 1)), dim3(CS_THREADS), 0, st, a);
  e = launch_scan(st, d_offsets, nq, d_offsets + nq, nullptr, 1, 0);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cell_search_kernel<2>, dim3(xcd_grid(HNC * CS_STRIPS, 1)), dim3(CS_THREADS), 0, st, a);
  return hipGetLastError();
}

// n_frames frames with identical set sizes; frame f reads a1 + f*a1_stride etc.  d_prune_ws holds
// match_pruned_workspace_bytes(nt, nq, n_frames) bytes (or is null: full scan).
hipError_t launch_match_batch(hipStream_t st, const float* d_a1, int n1, size_t a1_stride, const float* d_a2, int n2,
                              size_t a2_stride, float radius, int32_t* d_out_pairs, size_t out_stride, int* d_n_out,
                              unsigned long long* d_best, int* d_scratch, int n_cu, void* d_prune_ws, int n_frames,
                              int variant) {
  const int tree_is_1 = n1 >= n2;                 // vo_complete.cpp:15-20 (ties: a1 is the tree)
  const float* tree = tree_is_1 ? d_a1 : d_a2;
  const float* qry = tree_is_1 ? d_a2 : d_a1;
  const size_t ts = tree_is_1 ? a1_stride : a2_stride, qs = tree_is_1 ? a2_stride : a1_stride;
  const int nt = tree_is_1 ? n1 : n2, nq = tree_is_1 ? n2 : n1;
  const float r2 = radius * radius;
  const size_t best_stride = n_frames > 1 ? (size_t)nq : 0;
  const unsigned Z = (unsigned)n_frames;
  if (nq > 0 && nt > 0 && d_prune_ws && variant == 3) {
    hipError_t ep = launch_match_cells(st, tree, nt, qry, nq, radius, r2, d_best, d_prune_ws, n_frames, ts, qs,
                                       best_stride);
    if (ep != hipSuccess) return ep;
  } else if (nq > 0 && nt > 0 && d_prune_ws) {
    hipError_t ep = launch_match_pruned(st, tree, nt, qry, nq, radius, r2, d_best, d_prune_ws, n_cu, n_frames, ts, qs,
                                        best_stride);
    if (ep != hipSuccess) return ep;
  } else if (nq > 0) {
    hipLaunchKernelGGL(match_init_kernel, dim3((nq + 255) / 256, 1, Z), dim3(256), 0, st, d_best, nq, r2, best_stride);
    if (nt > 0) {
      const int qblocks = (nq + MB * QPT - 1) / (MB * QPT);
      // enough tree chunks to put ~8 workgroups on every CU, whole tiles each
      int want = (8 * (n_cu > 0 ? n_cu : 256) + qblocks * n_frames - 1) / (qblocks * n_frames);
      const int tiles = (nt + TILE - 1) / TILE;
      if (want > tiles) want = tiles;
      if (want < 1) want = 1;
      const int tiles_per_chunk = (tiles + want - 1) / want;
      const int chunk = tiles_per_chunk * TILE;
      const int nchunks = (nt + chunk - 1) / chunk;
      hipLaunchKernelGGL(match_kernel, dim3(qblocks, nchunks, Z), dim3(MB), 0, st, tree, nt, qry, nq, chunk, r2, d_best,
                         n_frames > 1 ? ts : 0, n_frames > 1 ? qs : 0, best_stride);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_match_compact(st, d_best, nq, tree_is_1, d_out_pairs, d_n_out, d_scratch, n_frames, best_stride,
                              out_stride);
}

hipError_t launch_match(hipStream_t st, const float* d_a1, int n1, const float* d_a2, int n2,
                        float radius, int32_t* d_out_pairs, int* d_n_out,
                        unsigned long long* d_best, int* d_scratch, int n_cu, void* d_prune_ws, int variant) {
  return launch_match_batch(st, d_a1, n1, 0, d_a2, n2, 0, radius, d_out_pairs, 0, d_n_out, d_best, d_scratch, n_cu,
                            d_prune_ws, 1, variant);
}

}  // namespace vo
