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
                                size_t out_stride, const int* d_n1, const int* d_n2, int cap1 = 0, int cap2 = 0,
                                const int* d_unres = nullptr);
extern const int SMALL_COMPACT;   // (geom.hip) up to this many items the compaction is one launch that reads every key: no frame is skipped

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

// RAGGED: frame z has its own set sizes d_n1[z] <= n1, d_n2[z] <= n2 (the arguments are then the capacities, i.e. the
// strides) and its own roles -- the larger set is the tree, a1 on ties (vo_complete.cpp:15-20) -- so `tree` / `qry` arrive
// as (a1, a2) and are told apart here.
template <bool RAGGED>
__global__ __launch_bounds__(MB) void match_kernel(const float* __restrict__ tree, int nt,
                                                   const float* __restrict__ qry, int nq,
                                                   int chunk, float r2,
                                                   unsigned long long* __restrict__ best, size_t tree_stride,
                                                   size_t qry_stride, size_t best_stride,
                                                   const int* __restrict__ d_n1, const int* __restrict__ d_n2) {
  tree += blockIdx.z * tree_stride; qry += blockIdx.z * qry_stride; best += blockIdx.z * best_stride;
  if (RAGGED) {
    int n1 = d_n1[blockIdx.z], n2 = d_n2[blockIdx.z];
    n1 = n1 < 0 ? 0 : (n1 > nt ? nt : n1); n2 = n2 < 0 ? 0 : (n2 > nq ? nq : n2);
    if (n1 >= n2) { nt = n1; nq = n2; }
    else { const float* t = tree; tree = qry; qry = t; nt = n2; nq = n1; }
    if ((int)(blockIdx.x * MB * QPT) >= nq || (int)(blockIdx.y * chunk) >= nt) return;     // (uniform: before any barrier)
  }
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
  const int* unres;     // hash-first (or null): per-frame count of queries the exact-duplicate pass left open; a frame with
                        //   none is skipped by every kernel, a query it has answered is not sorted in (best0 tells)
  const unsigned long long* best0;
};

// a key the exact-duplicate pass has written for an answered query: distance 0 and a tree index (an open query holds the
// "no hit" key: radius^2 and index 0xffffffff, like match_init_kernel writes)
__device__ __forceinline__ bool key_answered(unsigned long long k) { return (k >> 32) == 0ull && (unsigned)k != 0xffffffffu; }
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
  if (ms.unres && ms.unres[blockIdx.z] == 0) return;
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
#ifndef VO_SORT_BLOCKS
#define VO_SORT_BLOCKS 32
#endif
constexpr int SORT_BLOCKS = VO_SORT_BLOCKS;          // (one 50k frame's matcher: 16 / 24 / 32 / 64 / 128 workgroups -> 88 / 80 / 76 / 92 / 131 us)

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
  if (ms.unres && ms.unres[blockIdx.z] == 0) return;
  tree += blockIdx.z * ms.tree; qry += blockIdx.z * ms.qry;
  const unsigned long long* best_in = ms.unres ? ms.best0 + blockIdx.z * ms.best : nullptr;
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
    if (!is_t && best_in && key_answered(best_in[i - nt])) continue;
    const float* pt = is_t ? tree + 10 * (size_t)i : qry + 10 * (size_t)(i - nt);
    atomicAdd(&s_h[(is_t ? 0 : NBUCKET) + bucket_of(pt[bp.dimA], pt[bp.dimB], bp)], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * NBUCKET; k += 256) block_hist[(size_t)blockIdx.x * 2 * NBUCKET + k] = s_h[k];
}

// grid 2 (tree half, query half) x NBUCKET threads (one bin each)
__global__ __launch_bounds__(NBUCKET) void match_bucket_offsets_kernel(int* block_hist, int* starts, MatchStrides ms) {
  if (ms.unres && ms.unres[blockIdx.z] == 0) return;
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
  if (ms.unres && ms.unres[blockIdx.z] == 0) return;
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
    if (!is_t && ms.unres && key_answered(best[idx])) continue;       // answered by the exact-duplicate pass: not a query any more
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
  if (ms.unres && ms.unres[blockIdx.z] == 0) return;
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
                                      int n_frames, size_t tree_stride, size_t qry_stride, size_t best_stride,
                                      const int* d_unres = nullptr) {
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
  ms.unres = d_unres; ms.best0 = d_best;
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
// The tree is counting-sorted by cell -- coarse bin (c0, c1), numbered c0 * 20 + c1 whatever the grid, then fine bin
// (c2, c3) = c2 * nc3 + c3 inside it (<= 160 000 cells) -- in two levels, both on LDS histograms (a global atomic per
// point -- 64 scattered memory-side requests per wave -- was 2x dearer):
//   level 1 (cell_place_kernel, ONE pass over the input rows): the sets are cut into slices of CELL_SLICE points; a
//     workgroup reads its slice once, orders its records by the <= 400 coarse bins (c0, c1) in LDS and writes them back
//     as one contiguous run, together with a directory row (count, offset) per bin.  (Round 2 histogrammed the rows in a
//     first pass only to learn where every workgroup's records go, then read them again to place them: 0.19 ms of the
//     1.6 ms chain per 200 x 50k frames, bound by touching every 40-B row a second time.)
//   offsets (cell_offsets_kernel): column sums of the directory and their scan -> first slot of every coarse bin.
//   level 2 (cell_fine_kernel): one workgroup per group of FG consecutive coarse bins GATHERS their points from the slices'
//     runs (directory columns), orders them by (bin, fine bin) in LDS and writes the group's slice of the sorted tree and of
//     the start tables; the same workgroup copies the group's query records from their runs into one list per bin.
// A sorted tree point is 8 bytes: its original index and the FILTER WORD -- components 0..3 quantised to 8 bits each
// (see quant8 / the filter's proof below).  The search never reads a float of the tree before it has a survivor.
//
// Search: one workgroup per STRIP of CS_NB coarse bins (c0, c1 .. c1 + CS_NB - 1).  Every query of a bin can only meet
// tree points of the 3 x 3 coarse bins around it, i.e. nine CONTIGUOUS segments of the sorted tree (~140 points each
// on 50k uniform points) and nine rows of the start table: the workgroup stages the 3 x (CS_NB + 2) bins of its strip in
// LDS with coalesced loads (filter words, original indices, 16-bit relative starts) and every lane then visits its own
// <= 27 runs (3 x 3 x 3 cells in c0, c1, c2; contiguous along c3) out of LDS.
// A query visits, per component, the cells [cell(q - R), cell(q - R) + 2] clipped to cell(q + R): any tree value t
// with |t - q| < radius has cell(q - R) <= cell(t) (monotone) and (t - (q - R)) * scale < (r + R) * scale
// <= 1.9991 (1 + eps) < 2, so cell(t) <= cell(q - R) + 2.  A tree point outside that box differs from the
// query by at least the radius in one component, so that single non-negative term of the monotonically
// accumulated sum already reaches radius^2 and the strict test can never pass: the box changes no decision.
//
// The filter.  quant8(x, k) = clamp(floor((x - qlo_k) * qscale), 0, 255) is monotone in x, and for two floats a, b
// |quant8(a) - quant8(b)| <= |a - b| * qscale + 1.0001 (the floor costs one unit, the two float roundings of an unclamped
// value < 256 cost < 1e-4, clamping only shrinks a difference).  A tree point t that passes the decision has
// sum_k (t_k - q_k)^2 < r^2 (1 + 1e-6) over ALL components (the float sum is within 12 ulp of the real one), hence over
// components 0..3 sum_k |t_k - q_k| <= 2 * sqrt(sum_k (t_k - q_k)^2) < 2.000002 r, and the sum of absolute differences of
// the filter words is < 2.000002 r qscale + 4.0004 <= T = floor(2 R qscale) + 7.  So SAD(word_t, word_q) > T (ONE
// v_sad_u8 and a compare, against nine VALU instructions on a 16-byte float prefix in round 2) proves that t cannot be
// within the radius: the filter drops no point that could pass, and every survivor is decided from its full float row in
// the reference's operation order (brute_force_search.h:31-38), ties on the original index.  On U(-1,1)^10 appearances a
// random candidate survives with probability ~2e-4; a query meets ~27 candidates.
// Exactness never depends on the quantisation being useful: any qlo / qscale (0 when the sampled spread is empty or not
// finite) only changes how many survivors reach the decision.
//
// Segments too long for the LDS budget (clustered data) or a box reaching beyond the staged bins (cells narrower than R
// by a rounding) fall back to the same search on global memory.
constexpr int HK = 4;                    // hashed components
constexpr int HNC = 20;                  // cells per component, at most
constexpr int HCOARSE = HNC * HNC;       // coarse bins (c0, c1) / fine bins (c2, c3), at most
constexpr int HCPAD = 512;               // HCOARSE rounded up to a power of two (scan width)
constexpr int HROW = 408;                // u16 entries per row of the relative start table (>= HCOARSE + 1, 16-byte rows)
constexpr int CELL_SAMPLE = 2048;        // points per set that define the grid's bounds
#ifndef VO_CELL_SLICE
#define VO_CELL_SLICE 1792
#endif
constexpr int CELL_SLICE = VO_CELL_SLICE;         // points per level-1 workgroup: 7 per thread, ordered in 28 KiB of LDS
constexpr int CELL_PPT = CELL_SLICE / 256;
constexpr int CELL_MAX_SLICES = 1024;    // per set: the cell variant serves sets of up to 1 835 008 points (beyond: pruned scan)
#ifndef VO_CS_NB
#define VO_CS_NB 3
#endif
constexpr int CS_NB = VO_CS_NB;          // coarse bins per search workgroup (a strip along c1)
constexpr int CS_THREADS = CS_NB == 1 ? 192 : (CS_NB == 2 ? 320 : (CS_NB == 3 ? 448 : (CS_NB == 4 ? 576 : 704)));   // a coarse bin holds ~140 queries at 50k points
constexpr int CS_WG_PER_CU = CS_NB == 2 ? 5 : (CS_NB <= 3 ? 4 : (CS_NB == 4 ? 3 : 2));        // resident search workgroups per CU the registers are budgeted for
constexpr int CS_SURV = 2;               // further filter survivors a lane parks (beyond its first) before it decides them on the spot
constexpr int CS_CAP = CS_NB == 2 ? 2304 : (CS_NB <= 3 ? 2816 : (CS_NB == 4 ? 3328 : 3840));             // tree points a search workgroup can stage (8 B each; ~140 per bin at 50k)

bool match_cells_supported(int nt, int nq) {
  return (long long)nt <= (long long)CELL_MAX_SLICES * CELL_SLICE && (long long)nq <= (long long)CELL_MAX_SLICES * CELL_SLICE;
}

struct CellParams {
  int dim[HK];
  float lo[HK], scale[HK];
  int nc[HK];
  float R;
  float qlo[4], qscale;    // filter words: quant8 of components 0..3
  int T;                   // SAD threshold
};

// The grid of a frame from the per-component bounds of its sample: called by ONE WHOLE WAVE, lane k < 10 bringing the bounds of
// component k (the other lanes anything).  Everything is computed in every lane from values read across lanes -- the same
// choices, in the same order, as a scalar loop over the components would make (the four largest spans, ties to the lower
// component; the four largest among components 4..9 when none of them is flatter than half its counterpart) without one lane
// walking LDS arrays.
__device__ void make_cell_params_wave(float lo_k, float hi_k, float radius, CellParams* out) {
  const int lane = threadIdx.x & 63;
  const float sp0 = hi_k - lo_k;
  const float span = (lane < 10 && sp0 < INFINITY) ? sp0 : -1.f;      // empty / infinite / NaN ranges rank last
  // this component's place among all ten, and among components 4..9: descending span, ties to the lower index
  int rank_all = 0, rank_tail = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const float si = __shfl(span, i);
    const bool before = si > span || (si == span && i < lane);
    rank_all += before ? 1 : 0;
    rank_tail += (i >= 4 && before) ? 1 : 0;
  }
  int all4[HK], tail4[HK];
  bool tail_ok = true;
#pragma unroll
  for (int j = 0; j < HK; ++j) {
    all4[j] = __ffsll((unsigned long long)__ballot(lane < 10 && rank_all == j)) - 1;
    tail4[j] = __ffsll((unsigned long long)__ballot(lane >= 4 && lane < 10 && rank_tail == j)) - 1;
    tail_ok = tail_ok && __shfl(span, tail4[j]) >= 0.5f * __shfl(span, all4[j]);
  }
  CellParams cp;
  cp.R = radius * 1.001f;
#pragma unroll
  for (int j = 0; j < HK; ++j) {
    const int k = tail_ok ? tail4[j] : all4[j];
    const float sp = __shfl(span, k);
    int nc = 1;
    if (sp > 0.f && cp.R > 0.f) {
      const float f = sp / cp.R;
      nc = f >= (float)HNC ? HNC : (int)f;
      if (nc < 1) nc = 1;
    }
    cp.dim[j] = k; cp.lo[j] = __shfl(lo_k, k); cp.nc[j] = nc;
    cp.scale[j] = sp > 0.f ? (float)nc / sp : 0.f;       // cell width sp / nc >= R
  }
  // filter words: one scale for components 0..3 (the L1 bound needs a common unit), from their largest finite spread
  float wide = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float sk = __shfl(span, k);
    cp.qlo[k] = sk >= 0.f ? __shfl(lo_k, k) : 0.f;
    wide = sk > wide ? sk : wide;
  }
  cp.qscale = wide > 0.f ? 255.f / wide : 0.f;
  {
    const float t = 2.f * cp.R * cp.qscale;              // NaN / inf / huge -> every word passes (1020 is the largest SAD)
    cp.T = (t >= 0.f && t < 1000.f) ? (int)t + 7 : 1020;
  }
  if (lane == 0) *out = cp;
}

__device__ __forceinline__ unsigned quant8(float x, float lo, float scale) {
  float v = (x - lo) * scale;                            // monotone non-decreasing in x
  v = fminf(fmaxf(v, 0.f), 255.f);                       // NaN -> 0, +-inf clamp
  return (unsigned)v;                                    // truncation = floor (v >= 0)
}
__device__ __forceinline__ unsigned filter_word(float x0, float x1, float x2, float x3, const CellParams& cp) {
  return quant8(x0, cp.qlo[0], cp.qscale) | (quant8(x1, cp.qlo[1], cp.qscale) << 8) |
         (quant8(x2, cp.qlo[2], cp.qscale) << 16) | (quant8(x3, cp.qlo[3], cp.qscale) << 24);
}
__device__ __forceinline__ unsigned sad4(unsigned a, unsigned b) { return __builtin_amdgcn_sad_u8(a, b, 0u); }

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
  size_t cp, dir, t1, ql1, coarse_start, tree_word, tree_idx, q1, start_t, start_rel, open_start, open_rec, total;
};
#ifndef VO_OPEN_DIV
#define VO_OPEN_DIV 16
#endif
constexpr int OPEN_DIV_DEFAULT = VO_OPEN_DIV;   // the tree is streamed past a frame's open queries when they are <= 1 / 16 of its queries (and <= OPEN_MAX)
constexpr int OPEN_MAX = 2560;           // open queries per frame the tree can be streamed past (open_scan_kernel stages 9 records each in LDS)
constexpr int OPEN_RECS = 9 * OPEN_MAX;  //   a query is entered into every (c0, c1) column of its box
constexpr int OPEN_SROW_WS = 1024 * 8 + 8;   // (= OPEN_SROW below)
constexpr int SROW = HCOARSE + 1;        // ints per coarse bin in the absolute start table (its fine bins + its own end)
static int cell_slices(int n) { const int s = (n + CELL_SLICE - 1) / CELL_SLICE; return s < 1 ? 1 : s; }
static CellWs cell_ws_layout(int nt, int nq) {
  CellWs w;
  size_t o = 0;
  w.cp = o; o += align256(sizeof(CellParams));
  w.dir = o; o += align256(sizeof(unsigned) * (size_t)(cell_slices(nt) + cell_slices(nq)) * HCPAD);   // (count << 16 | offset) per (slice, bin)
  w.t1 = o; o += align256(sizeof(uint4) * (size_t)nt);                // level 1: (filter word, original index, fine | coarse << 16, -) in slice order
  w.ql1 = o; o += align256(sizeof(uint4) * (size_t)nq);               //   query records (below) in slice order
  w.coarse_start = o; o += align256(sizeof(int) * 2 * (HCPAD + 1));
  w.tree_word = o; o += align256(sizeof(unsigned) * (size_t)nt);      // sorted tree: filter word,
  w.tree_idx = o; o += align256(sizeof(int) * (size_t)nt);            //   original index
  w.q1 = o; o += align256(sizeof(uint4) * (size_t)nq);                // query records grouped by coarse bin: (original index, filter word,
                                                                      //   box = cell(q - R) per component : 5 bits each, width - 1 : 2 bits each, coarse bin)
  w.start_t = o; o += align256(sizeof(int) * (size_t)HCOARSE * SROW); // first tree slot of every cell, rows of SROW per coarse bin
  w.start_rel = o; o += align256(sizeof(unsigned short) * (size_t)HCOARSE * HROW);   // the same relative to the bin's first slot
                                                                      //   (16-bit, rows of HROW: what the search stages)
  // few open queries behind the exact-duplicate pass: their records ordered by coarse bin (open_collect_kernel -> open_scan_kernel)
  w.open_start = o; o += align256(sizeof(unsigned short) * OPEN_SROW_WS);       // 16-bit first slot of every bin (c0, c1, c2)
  w.open_rec = o; o += align256(sizeof(unsigned) * 2 * OPEN_RECS);              // filter words, then original indices
  w.total = o;
  return w;
}
size_t match_cells_workspace_bytes(int nt, int nq, int n_frames) {
  return (size_t)n_frames * cell_ws_layout(nt, nq).total;   // (ragged frames: call with nt = nq = the larger capacity)
}

struct CellArgs {
  const float* tree; const float* qry; int nt, nq;
  char* ws;                 // frame 0's block; frame f = ws + f * ws_stride
  CellWs w;
  size_t ws_stride, tree_stride, qry_stride, best_stride;
  int n_frames;
  int tb, qb;               // level-1 slices (workgroups) per frame over the tree / over the queries
  float radius, r2;
  unsigned long long* best;
  int* rs_offsets;          // radius search: [nq + 1] counts, then (after the scan) offsets
  int32_t* rs_indices;      // radius search: tree indices, room for rs_capacity
  int rs_capacity;
  const int* d_n1; const int* d_n2;   // ragged frames (or null): per-frame sizes of set 1 / set 2; tree / qry / nt / nq above are
                                      //   then set 1 / set 2 and their capacities, and every frame picks its roles (cell_sets)
  const int* unres;         // hash-first (or null): per-frame count of queries the exact-duplicate pass left open.  A frame with
                            //   none is skipped by every kernel; a query that pass has answered (key_answered(best[q])) is not
                            //   sorted in, so the search neither visits nor overwrites it
  int* todo;                // hash-first (or null): per-frame count of open queries left to the SORTED search -- unres[f], or 0 when
                            //   the frame's few open queries were settled by streaming the tree past them (open_scan_kernel)
  int open_div;             // that route is taken by a frame with at most min(OPEN_MAX, nq / open_div) open queries (0: never)
  const int* open_list;     // [n_frames][OPEN_MAX]: the open queries' indices as the lookup listed them
};

// the two sets of frame f with their roles: the larger one is the tree, set 1 on ties (vo_complete.cpp:15-20).  A: CellArgs
// or HashArgs -- (tree, qry, nt, nq) are (set 1, set 2, capacity 1, capacity 2) when the per-frame sizes are given.
template <class A>
__device__ __forceinline__ void cell_sets(const A& a, int f, const float*& tree, const float*& qry, int& nt, int& nq) {
  tree = a.tree + f * a.tree_stride; qry = a.qry + f * a.qry_stride; nt = a.nt; nq = a.nq;
  if (a.d_n1) {
    int n1 = a.d_n1[f], n2 = a.d_n2[f];
    n1 = n1 < 0 ? 0 : (n1 > a.nt ? a.nt : n1); n2 = n2 < 0 ? 0 : (n2 > a.nq ? a.nq : n2);
    if (n1 >= n2) { nt = n1; nq = n2; }
    else { const float* t = tree; tree = qry; qry = t; nt = n2; nq = n1; }
  }
}

// XCD-aware decomposition of a 1-D grid of 8 * ceil(n_frames / 8) * per_frame workgroups.  Workgroups are dealt
// round-robin over the 8 XCDs (observed; speed only, nothing depends on it), so giving every frame the workgroups of
// ONE residue class of blockIdx.x mod 8 keeps a frame's working set (4 MB of input, ~2 MB of records) in one
// XCD's L2: level 2 gathers what level 1 wrote there, segments staged by neighbouring strips hit there.
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
  coarse = c[0] * HNC + c[1];          // fixed stride: a bin's number -- hence where its starts live -- does not depend on the grid
  fine = c[2] * cp.nc[3] + c[3];
}
// a query's box, per component j: cell(q - R) in bits 5j..5j+4, (last cell - first cell) in bits 20+2j..21+2j
__device__ __forceinline__ unsigned query_box(const float* v, const CellParams& cp) {
  unsigned box = 0;
#pragma unroll
  for (int j = 0; j < HK; ++j) {
    const float x = pick10(v, cp.dim[j]);
    const int lo = cell_of(x - cp.R, cp.lo[j], cp.scale[j], cp.nc[j]);
    const int h = cell_of(x + cp.R, cp.lo[j], cp.scale[j], cp.nc[j]);
    const int hi = h < lo + 2 ? h : lo + 2;
    box |= ((unsigned)lo << (5 * j)) | ((unsigned)(hi - lo) << (20 + 2 * j));
  }
  return box;
}

// exclusive scan of HCPAD counters held two per thread (256 threads); returns the exclusive prefix of the thread's pair
// and the grand total.  s_w: 4 ints of LDS.
__device__ __forceinline__ int scan512(int c0, int c1, int* s_w, int& total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int v = c0 + c1;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int woff = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { const int c = s_w[w]; if (w < wave) woff += c; total += c; }
  return woff + incl - v;
}

// grid bounds: one workgroup per frame, min/max per component over a strided sample of both sets -> CellParams
__global__ __launch_bounds__(1024) void cell_bounds_kernel(CellArgs a) {
  const int f = blockIdx.x;
  if (a.unres && a.unres[f] == 0) return;
  const float* tree; const float* qry; int nt, nq;
  cell_sets(a, f, tree, qry, nt, nq);
  __shared__ float s_lo[16][10], s_hi[16][10];
  float lo[10], hi[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
  const int st_t = (nt + CELL_SAMPLE - 1) / CELL_SAMPLE, st_q = (nq + CELL_SAMPLE - 1) / CELL_SAMPLE;
  const int ns_t = st_t ? (nt + st_t - 1) / st_t : 0, ns_q = st_q ? (nq + st_q - 1) / st_q : 0;
  // <= 2 CELL_SAMPLE rows over 1024 threads: all of a thread's rows requested before the first is used
  constexpr int TRIPS = 2 * CELL_SAMPLE / 1024;
  // A row as 16 + 16 + 8 bytes (8-byte aligned: the hardware takes a 16-byte load at any dword).  4096 sampled rows per frame,
  // each in a line of its own: 21 MB fetched per 200 x 50k frames (FETCH_SIZE), 13 us in a counter pass -- 30 under the kernel
  // tracer, which is what requesting all rows at once, wider loads and the grid parameters computed across a wave instead of
  // by one lane were measured against (32.8 -> 30.2 us together).
  typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
  typedef float f2u __attribute__((ext_vector_type(2), aligned(8)));
  f4u va[TRIPS], vb[TRIPS];
  f2u vc[TRIPS];
  bool live[TRIPS];
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    const int i = threadIdx.x + t * 1024;
    live[t] = i < ns_t + ns_q;
    const int ic = live[t] ? i : 0;                        // (ns_t + ns_q > 0 whenever a frame gets here; row 0 of a set is valid)
    const float* row = (ic < ns_t || ns_q == 0) ? tree + 10 * (size_t)(ns_t ? ic : 0) * st_t : qry + 10 * (size_t)(ic - ns_t) * st_q;
    va[t] = *reinterpret_cast<const f4u*>(row); vb[t] = *reinterpret_cast<const f4u*>(row + 4); vc[t] = *reinterpret_cast<const f2u*>(row + 8);
  }
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    if (live[t]) {
      const float r[10] = {va[t].x, va[t].y, va[t].z, va[t].w, vb[t].x, vb[t].y, vb[t].z, vb[t].w, vc[t].x, vc[t].y};
#pragma unroll
      for (int k = 0; k < 10; ++k) { lo[k] = fminf(lo[k], r[k]); hi[k] = fmaxf(hi[k], r[k]); }          // fmin/fmax drop NaNs
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
  if (threadIdx.x < 64) {                                 // wave 0: lane k < 10 takes component k over the 16 waves, then the grid
    const int k = threadIdx.x < 10 ? threadIdx.x : 0;
    float l = INFINITY, h = -INFINITY;
#pragma unroll
    for (int w = 0; w < 16; ++w) { l = fminf(l, s_lo[w][k]); h = fmaxf(h, s_hi[w][k]); }
    make_cell_params_wave(l, h, a.radius, reinterpret_cast<CellParams*>(a.ws + f * a.ws_stride + a.w.cp));
  }
}

// level 1: one pass.  a.tb workgroups per frame take the tree's slices, a.qb the queries'.  A workgroup reads its <= 1792
// rows once (coalesced), ranks every point inside its coarse bin with an LDS atomic, scans the 512 counters, lays the
// records out by bin in LDS and copies them to the slice's own range of t1 / ql1 in that order: the stores are contiguous
// (round 2 wrote ~4 records into each of ~400 bins' runs).  The directory row tells level 2 where each bin's run starts.
__global__ __launch_bounds__(256) void cell_place_kernel(CellArgs a) {
  int f, blk;
  if (!xcd_frame_block(a.tb + a.qb, a.n_frames, f, blk)) return;
  if (a.unres && a.unres[f] == 0) return;
  char* ws = a.ws + f * a.ws_stride;
  const bool is_t = blk < a.tb;
  const float* tree; const float* qry; int nt, nq;
  cell_sets(a, f, tree, qry, nt, nq);
  const int n_set = is_t ? nt : nq;
  const int lo0 = (is_t ? blk : blk - a.tb) * CELL_SLICE;
  const int lo = lo0 < n_set ? lo0 : n_set;                // (ragged frames: a slice beyond the frame's set is empty)
  const int hi = lo + CELL_SLICE < n_set ? lo + CELL_SLICE : n_set;
  const float* src = is_t ? tree : qry;
  const unsigned long long* answered = (!is_t && a.unres) ? a.best + f * a.best_stride : nullptr;
  __shared__ int s_h[HCPAD];
  __shared__ int s_w[4];
  __shared__ uint4 s_rec[CELL_SLICE];
  const int tid = threadIdx.x;
  s_h[tid] = 0; s_h[tid + 256] = 0;
  const CellParams cp = *reinterpret_cast<const CellParams*>(ws + a.w.cp);
  __syncthreads();
  unsigned word[CELL_PPT], box[CELL_PPT];
  int bins[CELL_PPT], rank[CELL_PPT];
#pragma unroll
  for (int k = 0; k < CELL_PPT; ++k) {
    const int i = lo + k * 256 + tid;
    bins[k] = -1; word[k] = 0; rank[k] = 0; box[k] = 0;
    if (i < hi && !(answered && key_answered(answered[i]))) {
      float v10[10];
      load10(src + 10 * (size_t)i, v10);
      int coarse, fine;
      cell_bins(v10, cp, coarse, fine);
      bins[k] = fine | (coarse << 16);
      word[k] = filter_word(v10[0], v10[1], v10[2], v10[3], cp);
      if (!is_t) box[k] = query_box(v10, cp);
      rank[k] = atomicAdd(&s_h[coarse], 1);
    }
  }
  __syncthreads();
  const int c0 = s_h[2 * tid], c1 = s_h[2 * tid + 1];
  int total;
  const int ex = scan512(c0, c1, s_w, total);
  __syncthreads();                                        // every thread has read its counters
  s_h[2 * tid] = ex; s_h[2 * tid + 1] = ex + c0;          // first LDS slot of every bin
  {
    uint2* row = reinterpret_cast<uint2*>(reinterpret_cast<unsigned*>(ws + a.w.dir) + (size_t)blk * HCPAD);
    row[tid] = make_uint2(((unsigned)c0 << 16) | (unsigned)ex, ((unsigned)c1 << 16) | (unsigned)(ex + c0));
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CELL_PPT; ++k) {
    if (bins[k] >= 0) {
      const int slot = s_h[bins[k] >> 16] + rank[k];
      const int i = lo + k * 256 + tid;
      s_rec[slot] = is_t ? make_uint4(word[k], (unsigned)i, (unsigned)bins[k], 0u) : make_uint4((unsigned)i, word[k], box[k], (unsigned)(bins[k] >> 16));
    }
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(ws + (is_t ? a.w.t1 : a.w.ql1)) + lo;
  for (int j = tid; j < total; j += 256) dst[j] = s_rec[j];      // (total < hi - lo when answered queries were left out)
}

// offsets: grid (2 halves: tree, queries) x frames, HCPAD threads (one coarse bin each): column sums of the directory,
// exclusive scan over the bins -> coarse_start[2][HCPAD + 1]
__global__ __launch_bounds__(HCPAD) void cell_offsets_kernel(CellArgs a) {
  const int f = blockIdx.z;
  if (a.unres && a.unres[f] == 0) return;
  char* ws = a.ws + f * a.ws_stride;
  const unsigned* dir = reinterpret_cast<const unsigned*>(ws + a.w.dir);
  int* starts = reinterpret_cast<int*>(ws + a.w.coarse_start);
  __shared__ int s_w[HCPAD / 64];
  const int half = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b_lo = half ? a.tb : 0, b_hi = half ? a.tb + a.qb : a.tb;          // the rows that hold this set
  int v = 0;
#pragma unroll 4
  for (int b = b_lo; b < b_hi; ++b) v += (int)(dir[(size_t)b * HCPAD + tid] >> 16);
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < HCPAD / 64; ++w) { const int c = s_w[w]; if (w < wave) woff += c; tot += c; }
  starts[half * (HCPAD + 1) + tid] = woff + incl - v;
  if (tid == 0) starts[half * (HCPAD + 1) + HCPAD] = tot;
}

// level 2: one workgroup per GROUP of FG consecutive coarse bins (c0, c1 .. c1 + FG - 1).  A slice's records are ordered by
// coarse bin, so the group's points form ONE run per slice (directory columns coarse .. coarse + FG - 1): gather them,
// counting-sort them by (bin, fine bin) in LDS, write the group's slice of the sorted tree (filter word, original index)
// and of the start tables.  The group's query records are copied from their runs into one contiguous list.
// The work of a bin is tiny and its time is barriers and DEPENDENT memory round trips, hence the grouping (FG bins share
// every barrier and every round trip), and everything whose address is known up front (the grid, the group's starts, both
// directory column groups) is requested at once; both gathers (tree records, query records) fly together.
// The absolute start table (start_t) is what the search falls back to when a bin holds >= 32768 points (the 16-bit table
// cannot express its slots): written for such bins only.  Bit 15 of a 16-bit entry says "a run of <= 3 cells starting
// here holds more than 4 points": the search's first pass looks at four candidates per run and leaves such runs to its
// second pass.
#ifndef VO_FG
#define VO_FG 5
#endif
constexpr int FG = VO_FG;                // coarse bins per level-2 workgroup (HNC is a multiple)
constexpr int FINE_PPT = FG <= 5 ? 4 : 8;   // tree records per thread kept in registers between counting and placing
constexpr int FINE_QPT = FG <= 5 ? 4 : 8;   // query records per thread
constexpr int FCNT = FG <= 5 ? 2048 : 4096; // counters: FG * n_fine <= FG * 400
static_assert(HNC % FG == 0 && FG * HCOARSE <= FCNT, "");
__global__ __launch_bounds__(256) void cell_fine_kernel(CellArgs a) {
  int f, grp;
  if (!xcd_frame_block(HCOARSE / FG, a.n_frames, f, grp)) return;
  if (a.unres && a.unres[f] == 0) return;
  char* ws = a.ws + f * a.ws_stride;
  const int coarse = grp * FG;                             // first bin of the group (same c0 for all: HNC % FG == 0)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned* dir = reinterpret_cast<const unsigned*>(ws + a.w.dir);
  const int* cstart = reinterpret_cast<const int*>(ws + a.w.coarse_start);
  // round trip 1
  auto dir_entry = [&](int row) {                           // the group's run in slice `row`: (points, first slot in the slice)
    const unsigned* p = dir + (size_t)row * HCPAD + coarse;
    unsigned cnt = 0;
#pragma unroll
    for (int k = 0; k < FG; ++k) cnt += p[k] >> 16;
    return make_uint2(cnt, p[0] & 0xffffu);
  };
  const uint2 e_t0 = tid < a.tb ? dir_entry(tid) : make_uint2(0u, 0u);
  const uint2 e_q0 = tid < a.qb ? dir_entry(a.tb + tid) : make_uint2(0u, 0u);
  int bstart[FG + 1];
#pragma unroll
  for (int k = 0; k <= FG; ++k) bstart[k] = cstart[coarse + k];
  const int begin = bstart[0], end = bstart[FG], n = end - begin;
  int qstart[FG + 1];
#pragma unroll
  for (int k = 0; k <= FG; ++k) qstart[k] = cstart[HCPAD + 1 + coarse + k];
  const int qn = qstart[FG] - qstart[0];
  const CellParams cp = *reinterpret_cast<const CellParams*>(ws + a.w.cp);
  const int n_fine = cp.nc[2] * cp.nc[3];
  const int c1_first = coarse % HNC;
  if (coarse / HNC >= cp.nc[0] || c1_first >= cp.nc[1]) return;         // no bin of this grid in the group (it holds nothing)
  const int bins_here = cp.nc[1] - c1_first < FG ? cp.nc[1] - c1_first : FG;
  __shared__ int s_cnt[FCNT];
  __shared__ int s_w[4];
  __shared__ int s_rs[2][CELL_MAX_SLICES + 1], s_src[2][CELL_MAX_SLICES];   // per set and run: first element number, first source slot
  __shared__ int s_carry;
  __shared__ int s_qcur[FG];
#pragma unroll
  for (int k = 0; k < FCNT / 256; ++k) s_cnt[tid + 256 * k] = 0;
  if (tid < FG) {
    int v = qstart[0];
#pragma unroll
    for (int k = 1; k < FG; ++k) if (tid == k) v = qstart[k];
    s_qcur[tid] = v;                                      // cursor of every bin's query list
  }
  // the runs of one set: rows [r0, r0 + nr) of the directory (the first 256 rows were requested above)
  auto build_runs = [&](int set, int r0, int nr, uint2 e_first) {
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nr; b0 += 256) {
      const int t = b0 + tid;
      uint2 e = e_first;
      if (b0 > 0) e = t < nr ? dir_entry(r0 + t) : make_uint2(0u, 0u);
      const int cnt = (int)e.x;
      int incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
      if (lane == 63) s_w[wave] = incl;
      __syncthreads();
      int woff = s_carry, tot = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { const int c = s_w[w]; if (w < wave) woff += c; tot += c; }
      if (t < nr) { s_rs[set][t] = woff + incl - cnt; s_src[set][t] = t * CELL_SLICE + (int)e.y; }
      __syncthreads();
      if (tid == 0) s_carry += tot;
      __syncthreads();
    }
    if (tid == 0) s_rs[set][nr] = s_carry;
    __syncthreads();
  };
  // source slot of element i: the last run that starts at or before i (empty runs share a start with their successor and
  // are skipped by "last")
  auto source_of = [&](int set, int i, int nr) {
    int lo = 0, hi = nr;                                  // invariant: s_rs[lo] <= i < s_rs[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_rs[set][mid] <= i) lo = mid; else hi = mid;
    }
    return s_src[set][lo] + (i - s_rs[set][lo]);
  };
  build_runs(0, 0, a.tb, e_t0);
  build_runs(1, a.tb, a.qb, e_q0);
  const uint4* t1 = reinterpret_cast<const uint4*>(ws + a.w.t1);
  const uint4* ql1 = reinterpret_cast<const uint4*>(ws + a.w.ql1);
  uint4* q1 = reinterpret_cast<uint4*>(ws + a.w.q1);
  unsigned* tree_word = reinterpret_cast<unsigned*>(ws + a.w.tree_word);
  int* tree_idx = reinterpret_cast<int*>(ws + a.w.tree_idx);
  const bool in_regs = n <= 256 * FINE_PPT, q_in_regs = qn <= 256 * FINE_QPT;
  // counter of a record: (its bin's place in the group) * n_fine + its fine bin
  auto counter_of = [&](unsigned bins) { return (int)((bins >> 16) - (unsigned)coarse) * n_fine + (int)(bins & 0xffffu); };
  // round trip 2: both gathers
  uint4 rec[FINE_PPT], qrec[FINE_QPT];
  int rk[FINE_PPT];
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < FINE_PPT; ++k) {
      const int i = tid + 256 * k;
      rec[k] = make_uint4(0u, 0u, 0u, 0u);
      if (i < n) rec[k] = t1[source_of(0, i, a.tb)];
    }
  }
  if (q_in_regs) {
#pragma unroll
    for (int k = 0; k < FINE_QPT; ++k) {
      const int i = tid + 256 * k;
      qrec[k] = make_uint4(0u, 0u, 0u, 0u);
      if (i < qn) qrec[k] = ql1[source_of(1, i, a.qb)];
    }
  }
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < FINE_PPT; ++k) {
      rk[k] = 0;
      if (tid + 256 * k < n) rk[k] = atomicAdd(&s_cnt[counter_of(rec[k].z)], 1);
    }
  } else {
    for (int i = tid; i < n; i += 256) atomicAdd(&s_cnt[counter_of(t1[source_of(0, i, a.tb)].z)], 1);
  }
  __syncthreads();
  // exclusive scan of the FCNT counters, eight consecutive ones per thread
  int c[FCNT / 256];
  int sum = 0;
#pragma unroll
  for (int k = 0; k < FCNT / 256; ++k) { c[k] = s_cnt[(FCNT / 256) * tid + k]; sum += c[k]; }
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int run = begin + incl - sum;
#pragma unroll
  for (int w = 0; w < 4; ++w) if (w < wave) run += s_w[w];
#pragma unroll
  for (int k = 0; k < FCNT / 256; ++k) { s_cnt[(FCNT / 256) * tid + k] = run; run += c[k]; }   // cursors = absolute first slots
  __syncthreads();
  // the start tables of the group's bins.  Entry `fine` of bin b: first slot of cell (b, fine); entry n_fine: the bin's end.
  for (int i = tid; i < bins_here * (n_fine + 1); i += 256) {
    const int b = i / (n_fine + 1), fine = i - b * (n_fine + 1);
    int b_begin = bstart[0], b_end = bstart[1];
#pragma unroll
    for (int k = 1; k < FG; ++k) if (b == k) { b_begin = bstart[k]; b_end = bstart[k + 1]; }
    const int at = fine < n_fine ? s_cnt[b * n_fine + fine] : b_end;
    const int ahead_i = fine + 3 < n_fine ? fine + 3 : n_fine;
    const int ahead = ahead_i < n_fine ? s_cnt[b * n_fine + ahead_i] : b_end;
    const int r = at - b_begin;
    const bool big = b_end - b_begin >= 32768;
    unsigned short* rel = reinterpret_cast<unsigned short*>(ws + a.w.start_rel) + (size_t)(coarse + b) * HROW;
    rel[fine] = (unsigned short)(big ? 0x7fff : (r | (ahead - at > 4 ? 0x8000 : 0)));
    if (big) reinterpret_cast<int*>(ws + a.w.start_t)[(size_t)(coarse + b) * SROW + fine] = at;
  }
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < FINE_PPT; ++k) {
      const int i = tid + 256 * k;
      if (i < n) {
        const int pos = s_cnt[counter_of(rec[k].z)] + rk[k];
        tree_word[pos] = rec[k].x;
        tree_idx[pos] = (int)rec[k].y;
      }
    }
  } else {
    for (int i = tid; i < n; i += 256) {
      const uint4 r = t1[source_of(0, i, a.tb)];
      const int pos = atomicAdd(&s_cnt[counter_of(r.z)], 1);
      tree_word[pos] = r.x;
      tree_idx[pos] = (int)r.y;
    }
  }
  // the group's queries, every bin's list contiguous (the order inside a bin's list is immaterial)
  if (q_in_regs) {
#pragma unroll
    for (int k = 0; k < FINE_QPT; ++k) {
      const int i = tid + 256 * k;
      if (i < qn) q1[atomicAdd(&s_qcur[qrec[k].w - (unsigned)coarse], 1)] = qrec[k];
    }
  } else {
    for (int i = tid; i < qn; i += 256) {
      const uint4 r = ql1[source_of(1, i, a.qb)];
      q1[atomicAdd(&s_qcur[r.w - (unsigned)coarse], 1)] = r;
    }
  }
}

// MODE 0: best match per query (bestMatchFull);  MODE 1 / 2: count / write ALL tree points with d2 < r2
// (fullSearch, eigen_kdtree.h:56-71 + bruteForceSearch, brute_force_search.h:3-20)
// One workgroup serves a STRIP of CS_NB coarse bins (c0, c1f .. c1f + CS_NB - 1): their queries are contiguous in
// q1_idx, and their neighbourhoods overlap -- 3 x (CS_NB + 2) staged bins instead of 9 per bin.
constexpr int CS_PLANES = 3 * (CS_NB + 2);
constexpr int CS_STRIPS = (HNC + CS_NB - 1) / CS_NB;     // strips per c0 row, at most
// a 40-byte appearance row (8-byte aligned) as three loads instead of five: every load instruction of a gather costs the
// texture path one request per lane, whatever its width (global loads take any 4-byte alignment on gfx950)
struct __attribute__((packed, aligned(8))) Row10 { float v[10]; };
__device__ __forceinline__ Row10 load_row(const float* p) {
  Row10 r;
  typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
  typedef float f2u __attribute__((ext_vector_type(2), aligned(8)));
  const f4u a = *reinterpret_cast<const f4u*>(p), b = *reinterpret_cast<const f4u*>(p + 4);
  const f2u c = *reinterpret_cast<const f2u*>(p + 8);
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  r.v[8] = c.x; r.v[9] = c.y;
  return r;
}

template <int MODE>
__global__ __launch_bounds__(CS_THREADS, CS_THREADS * CS_WG_PER_CU / 256) void cell_search_kernel(CellArgs a) {
  int f, blk;
  if (!xcd_frame_block(HNC * CS_STRIPS, a.n_frames, f, blk)) return;
  if (a.unres && a.unres[f] == 0) return;
  char* ws = a.ws + f * a.ws_stride;
  const int c0 = blk / CS_STRIPS, c1f = (blk - c0 * CS_STRIPS) * CS_NB;
  const int nb_here = HNC - c1f < CS_NB ? HNC - c1f : CS_NB;
  const int coarse0 = c0 * HNC + c1f;                     // bins are numbered c0 * HNC + c1 whatever the grid: nothing below
  const int* __restrict__ cstart_t = reinterpret_cast<const int*>(ws + a.w.coarse_start);   // waits for the grid's parameters
  const int* __restrict__ cstart_q = cstart_t + (HCPAD + 1);
  const int tid = threadIdx.x;
  __shared__ unsigned s_word[CS_CAP + 4];                 // filter words of the staged tree points (+4: a visit reads four slots)
  __shared__ int s_idx[CS_CAP];                           // their original indices
  __shared__ __attribute__((aligned(16))) unsigned short s_start[CS_PLANES][HROW];   // cell starts of the staged bins, relative to each bin's first slot
  __shared__ int s_gbase[CS_PLANES], s_lbase[CS_PLANES + 1], s_len[CS_PLANES];   // per bin: first slot in the sorted tree, in LDS, length
  __shared__ int s_surv[CS_SURV][CS_THREADS];             // parked filter survivors (original indices), lane-private columns
  // round trip 1: the strip's queries, the staged bins' segments, the grid -- all addressed by the block index alone.
  // staged bin `s` = (row, col): coarse bin (c0 - 1 + row, c1f - 1 + col); bins outside the grid have length 0
  if (tid < CS_PLANES) {
    const int p0 = c0 - 1 + tid / (CS_NB + 2), p1 = c1f - 1 + tid % (CS_NB + 2);
    int gb = 0, len = 0;
    if (p0 >= 0 && p0 < HNC && p1 >= 0 && p1 < HNC) { gb = cstart_t[p0 * HNC + p1]; len = cstart_t[p0 * HNC + p1 + 1] - gb; }
    s_gbase[tid] = gb; s_len[tid] = len;
  }
  const int qb = cstart_q[coarse0], qe = cstart_q[coarse0 + nb_here];
  const CellParams cp = *reinterpret_cast<const CellParams*>(ws + a.w.cp);
  if (qb >= qe) return;                                   // no query lives here: nothing to stage
  const int n2 = cp.nc[2], n3 = cp.nc[3];
  const int n_fine = n2 * n3;
  const int* __restrict__ start_t = reinterpret_cast<const int*>(ws + a.w.start_t);
  const unsigned* __restrict__ tree_word = reinterpret_cast<const unsigned*>(ws + a.w.tree_word);
  const int* __restrict__ tree_idx = reinterpret_cast<const int*>(ws + a.w.tree_idx);
  const uint4* __restrict__ q1 = reinterpret_cast<const uint4*>(ws + a.w.q1);
  const float* tree; const float* qry;
  { int nt_f, nq_f; cell_sets(a, f, tree, qry, nt_f, nq_f); }
  unsigned long long* best = a.best + f * a.best_stride;
  // round trip 2: the first batch of query records (original index, filter word, box: written by level 1, which had the
  // row in registers anyway) beside the staging loads.  The query's floats are not touched before a survivor is decided.
  int qi = qb + tid;
  uint4 qrec = q1[qi < qe ? qi : qb];
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
  const bool staged = total <= CS_CAP && longest < 32768;
  __syncthreads();
  if (staged) {
    // flattened over the staged bins: every thread's loads are independent, several in flight at once
#pragma unroll 4
    for (int i = tid; i < total; i += CS_THREADS) {
      int s = 0;
#pragma unroll
      for (int j = 1; j < CS_PLANES; ++j) s += (i >= s_lbase[j]) ? 1 : 0;   // empty bins share a base: the last one wins,
      const int g = s_gbase[s] + (i - s_lbase[s]);                          // and only a bin with points can own slot i
      s_word[i] = tree_word[g];
      s_idx[i] = tree_idx[g];
    }
    if (tid < 4) s_word[total + tid] = 0u;                // the read-ahead behind the last staged point
    // the bins' rows of the 16-bit start table, eight entries per load; an empty bin's row reads as zeros (all its runs
    // are empty), so the visits below need no test for it
    const uint4* rel = reinterpret_cast<const uint4*>(ws + a.w.start_rel);
    const int vec_per_row = (n_fine + 1 + 7) / 8;
#pragma unroll 2
    for (int i = tid; i < CS_PLANES * (HROW / 8); i += CS_THREADS) {
      const int s = i / (HROW / 8), v = i - s * (HROW / 8);
      if (v < vec_per_row) {
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (s_len[s] > 0) {
          const int cb = (c0 - 1 + s / (CS_NB + 2)) * HNC + (c1f - 1 + s % (CS_NB + 2));
          r = rel[(size_t)cb * (HROW / 8) + v];
        }
        reinterpret_cast<uint4*>(&s_start[s][0])[v] = r;
      }
    }
  }
  __syncthreads();

  const unsigned T = (unsigned)cp.T;
#if defined(VO_CS_EXP) && VO_CS_EXP == 1
  if (a.r2 >= 0.f) return;                                // experiment: staging only
#endif
  for (; qi < qe; qi += CS_THREADS) {
    if (qi >= qb + CS_THREADS) qrec = q1[qi];             // later batches (rare)
    const int qorig = (int)qrec.x;
    const unsigned qw = qrec.y;
    int c_lo[HK], c_hi[HK];
#pragma unroll
    for (int j = 0; j < HK; ++j) {
      c_lo[j] = (int)((qrec.z >> (5 * j)) & 31u);
      c_hi[j] = c_lo[j] + (int)((qrec.z >> (20 + 2 * j)) & 3u);
    }
    const float* qrow = qry + 10 * (size_t)qorig;
    float bd = a.r2;
    int bi = -1;
    int n_hit = 0;
    int out_at = 0;
    if (MODE == 2) out_at = a.rs_offsets[qorig];
    // the decision itself: the reference's unfused left-to-right sum (brute_force_search.h:34) over the whole row
    auto decide = [&](const int ti, const Row10& t, const Row10& q) {
      float d = t.v[0] - q.v[0];
      float s = d * d;
#pragma unroll
      for (int k = 1; k < 10; ++k) { d = t.v[k] - q.v[k]; s += d * d; }
      if (MODE == 0) {
        if (s < bd || (s == bd && bi >= 0 && ti < bi)) { bd = s; bi = ti; }
      } else if (s < bd) {                               // bd stays radius^2: every point inside the ball
        if (MODE == 2 && out_at + n_hit < a.rs_capacity) a.rs_indices[out_at + n_hit] = ti;
        ++n_hit;
      }
    };
    auto consider = [&](const unsigned w, const int ti, const Row10& q) {  // filter, then fetch the row and decide on the spot
      if (sad4(w, qw) <= T) decide(ti, load_row(tree + 10 * (size_t)ti), q);
    };
    // the query's box in staged-bin coordinates
    const int row_lo = c_lo[0] - (c0 - 1), row_hi = c_hi[0] - (c0 - 1);
    const int col_lo = c_lo[1] - (c1f - 1), col_hi = c_hi[1] - (c1f - 1);
    if (staged && row_lo >= 0 && row_hi <= 2 && col_lo >= 0 && col_hi <= CS_NB + 1) {
      // Phase 1, every lane in lockstep: visit the 27 runs (bin row i0, bin column i1, c2 = c_lo[2] + j; a run is
      // contiguous along c3), FOUR candidates per visit (a run holds ~1.1 points): four filter words from LDS, four
      // v_sad_u8, one minimum, one compare -- and one bit per run that holds a survivor.  Nothing is masked here: an index
      // beyond the box is clamped onto its last cell (a run visited twice), and slots behind a short run belong to the next
      // cells of the sorted tree; neither can produce a wrong result (a flagged run is re-examined below with exact bounds,
      // and a point outside the box can never pass the decision).  Runs longer than four points take extra trips.
      const int d2 = c_hi[2] - c_lo[2], d3 = c_hi[3] - c_lo[3];
      int ia[3];                                          // byte offset of the run's first table entry inside a bin's row
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c2 = c_lo[2] + (j < d2 ? j : d2);
        ia[j] = 2 * (c2 * n3 + c_lo[3]);
      }
      unsigned hits = 0;
      int sbin[3], scol[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        sbin[i] = (row_lo + i < row_hi ? row_lo + i : row_hi) * (CS_NB + 2);
        scol[i] = col_lo + i < col_hi ? col_lo + i : col_hi;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(ia[j]));    // keep them: re-deriving one costs a v_mul_lo_u32 per run
#pragma unroll
      for (int i0 = 0; i0 < 3; ++i0)
#pragma unroll
        for (int i1 = 0; i1 < 3; ++i1) {
          const int s = sbin[i0] + scol[i1];
          const int lb = s_lbase[s];
          const char* trow = reinterpret_cast<const char*>(&s_start[0][0]) + __umul24((unsigned)s, (unsigned)(HROW * 2));
          // the bin's three runs: ONE table entry each (first slot; bit 15: the run may hold more than four points -- then
          // the second pass looks at all of it), four filter words, no branch
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const unsigned e = *reinterpret_cast<const unsigned short*>(trow + ia[j]);
            const unsigned* w = &s_word[lb + (int)(e & 0x7fffu)];
            unsigned m = sad4(w[0], qw);
            { const unsigned m1 = sad4(w[1], qw), m2 = sad4(w[2], qw), m3 = sad4(w[3], qw);
              m = m < m1 ? m : m1; m = m < m2 ? m : m2; m = m < m3 ? m : m3; }
            m = e >= 0x8000u ? 0u : m;
            hits |= m <= T ? (1u << (9 * i0 + 3 * i1 + j)) : 0u;   // (an empty run flagged by the slots behind it costs one empty visit below)
          }
        }
      // Phase 2: the flagged runs again, with exact bounds.  A survivor needs its row from global memory (a miss of ~1-2 us)
      // and the query's: the query's row is requested before the runs are walked, the first survivor's row the moment it is
      // found -- both fly while the remaining runs are examined -- and further survivors of MODE 0 are parked (their original
      // index) and fetched together afterwards; the other modes decide on the spot.
      const Row10 qv = load_row(qrow);
      Row10 first_row;
      int first_ti = -1;
      int n_surv = 0;
#if defined(VO_CS_EXP) && VO_CS_EXP == 2
      if (a.r2 >= 0.f) hits = hits & 0x80000000u;         // experiment: no survivor is examined
#endif
      while (hits != 0u) {
        const int bit = __builtin_ctz(hits);
        hits &= hits - 1u;
        const int i0 = bit / 9, i1 = (bit - 9 * i0) / 3, j = bit - 9 * i0 - 3 * i1;
        if (row_lo + i0 > row_hi || col_lo + i1 > col_hi || j > d2) continue;     // a clamped duplicate
        const int s = (row_lo + i0) * (CS_NB + 2) + col_lo + i1;
        const int lb = s_lbase[s];
        const int e0 = (c_lo[2] + j) * n3 + c_lo[3];
        const int sa = s_start[s][e0] & 0x7fff, sb = s_start[s][e0 + d3 + 1] & 0x7fff;
        for (int k = sa; k < sb; ++k) {
          const unsigned w = s_word[lb + k];
          if (sad4(w, qw) <= T) {
            const int ti = s_idx[lb + k];
            if (MODE == 0 && first_ti < 0) { first_ti = ti; first_row = load_row(tree + 10 * (size_t)ti); }
            else if (MODE == 0 && n_surv < CS_SURV) { s_surv[n_surv][tid] = ti; ++n_surv; }
            else decide(ti, load_row(tree + 10 * (size_t)ti), qv);
          }
        }
      }
      if (MODE == 0) {
        if (first_ti >= 0) decide(first_ti, first_row, qv);
        for (int k = 0; k < n_surv; k += 2) {             // two more rows per lane in flight
          const bool h1 = k + 1 < n_surv;
          const int ia_ = s_surv[k][tid], ib_ = s_surv[h1 ? k + 1 : k][tid];
          const Row10 ra = load_row(tree + 10 * (size_t)ia_), rb = load_row(tree + 10 * (size_t)ib_);
          decide(ia_, ra, qv);
          if (h1) decide(ib_, rb, qv);
        }
      }
    } else {
      // the same search on global memory (segments beyond the LDS budget, or a box beyond the staged bins)
      const unsigned short* rel_g = reinterpret_cast<const unsigned short*>(ws + a.w.start_rel);
      const Row10 qv = load_row(qrow);
      for (int x0 = c_lo[0]; x0 <= c_hi[0]; ++x0)
        for (int x1 = c_lo[1]; x1 <= c_hi[1]; ++x1) {
          const int cb = x0 * HNC + x1;
          const int b_begin = cstart_t[cb];
          const bool big = cstart_t[cb + 1] - b_begin >= 32768;          // its slots do not fit the 16-bit table
          for (int x2 = c_lo[2]; x2 <= c_hi[2]; ++x2) {
            const int k0 = x2 * n3 + c_lo[3], k1 = x2 * n3 + c_hi[3] + 1;
            const int p0 = big ? start_t[(size_t)cb * SROW + k0] : b_begin + (rel_g[(size_t)cb * HROW + k0] & 0x7fff);
            const int p1 = big ? start_t[(size_t)cb * SROW + k1] : b_begin + (rel_g[(size_t)cb * HROW + k1] & 0x7fff);
            for (int p = p0; p < p1; ++p) consider(tree_word[p], tree_idx[p], qv);
          }
        }
    }
    if (MODE == 0)
      best[qorig] = bi >= 0 ? (((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)(unsigned)bi)
                            : (((unsigned long long)__float_as_uint(a.r2) << 32) | 0xffffffffull);
    if (MODE == 1) a.rs_offsets[qorig] = n_hit;
  }
}


// ---- exact duplicates first ("hash-first") ------------------------------------------------------------------------
// The appearance of a landmark is COPIED from frame to frame (the reference's dataset: max |difference| 0 over all frames,
// SURVEY appendix C; the synthetic generators likewise), so for almost every query the nearest tree point is a bitwise copy
// of it, at distance exactly 0 -- the smallest value the reference's sum of squares (brute_force_search.h:34) can take.  This
// pass finds those copies by hashing, exactly:
//   * a row is SAFE when its ten components are finite and |x| >= 2^-40.  For a safe query q and ANY tree row t,
//     d2(t, q) == 0 in the reference's float arithmetic iff t == q component by component: a difference t_k - q_k != 0
//     has magnitude >= ulp(2^-40)/2 = 2^-64, its square 2^-128 is a normal float, and a sum of non-negative terms with one
//     positive term is positive.  (Without the bound tiny differences square to 0 by underflow: such rows go to the
//     general search, like NaN / inf rows, which no distance test can pass.)
//   * the frame's table is cut into P parts BY HASH (top bits), S 32-bit words each; a streaming map hashes every tree row once
//     and queues (hash, index) records per part, one workgroup per part takes its queue and
//     inserts them into an open-addressing table in LDS (word = tag | index, index in the low `ib` bits; linear probing from
//     an even home slot; a row equal to one already present takes the SMALLER index into that entry by atomicMin after
//     comparing the two rows, so the table holds one entry per distinct row -- the lowest index, which is the tie rule of this
//     library's matcher), then stores the finished part as one contiguous image.  No global atomics (level 1's comment above:
//     a global atomic per point is 64 scattered memory-side requests per wave).
//   * a safe query reads ONE 16-byte window -- the home slot and the three behind it -- of the part its hash selects (a first version
//     cut the tree by position and looked into every slice's table: 8 scattered loads per query at ~200 G scattered
//     lane-loads/s chip-wide made the lookup 0.64 ms per 200 x 50k frames), fetches the row of an entry whose tag agrees
//     and compares all ten floats: equal -> key (distance 0, that entry's index).  With radius^2 > 0 the key is FINAL: no tree
//     point is closer than 0, and every tree point at distance 0 is a bitwise copy, merged into that entry.
// Queries without a copy (new landmarks, noise, unsafe rows; rows a full part could not take) keep the "no hit" key and are
// counted per frame; the general search that follows (bucket-pruned scan or cell-hash search) then sorts in only those queries
// and is skipped altogether by a frame that has none.  Results are identical to the general search alone for every input
// (tests/test_gpu_parity.py::test_matcher_variants_agree, tests/test_gpu_hashfirst.py).
constexpr unsigned HJ_EMPTY = 0xffffffffu;
#ifndef VO_HJ_THREADS
#define VO_HJ_THREADS 768
#endif
constexpr int HJ_THREADS = VO_HJ_THREADS;           // 12 waves (two workgroups of the table step per CU at 2^14-word parts)
constexpr int HJ_NW = HJ_THREADS / 64;
constexpr int HJ_STRIP = 320;                       // float2 per strip: 64 rows of 40 bytes

struct HashArgs {
  const float* tree; const float* qry; int nt, nq;          // as CellArgs (set 1 / set 2 and capacities when ragged)
  size_t tree_stride, qry_stride, best_stride;
  const int* d_n1; const int* d_n2;
  unsigned* tables;           // frame 0's table: parts << log2s words; frame f = tables + f * tables_stride
  size_t tables_stride;
  uint2* queues;              // [n_frames][parts][qcap] records (hash, tree index): step 1 -> step 2
  int* cursors;               // [n_frames][HJ_MAXP] records per part (zeroed per call, beside unres)
  int qcap;                   // records a part's queue holds
  int* unres;                 // [n_frames] queries left open (zeroed per call)
  unsigned* sample_mask;      // [n_frames] (zeroed per call) bit k: sample query k of the frame has a bitwise copy in the tree -- a frame
                              //   with fewer than sample_min of its HJ_SAMPLES samples found skips the tables and the lookup: its queries all go to the search
  int sample_min;             // automatic mode: HJ_SAMPLE_MIN (the pass pays while fewer than ~10 % of a frame's queries lack a copy);
                              //   forced modes 4 / 5: 1 (any copy at all)
  int* open_list;             // [n_frames][OPEN_MAX] their indices when there are at most OPEN_MAX, in no particular order (hash_open_kernel)
  size_t queue_stride;        // uint2 per frame of `queues`; the lookup reuses a frame's queues for its workgroups' open queries:
  int seg_counts;             //   ints [0, seg_counts) the segments (256 HJ_Q per workgroup), behind them one count per workgroup
  unsigned long long* best;
  int32_t* out_pairs;         // the call's pair output (or null): the lookup writes pair j of a frame at slot j -- the final
  size_t out_stride;          //   place when every query of the frame finds its copy (then the compaction has nothing to do)
  int tree_is_1;              // roles of a call with one size for all frames (ragged frames: from the sizes)
  float r2;
  int n_frames, log2p, ib, qblocks, hblocks;   // parts = 1 << log2p; ib: index bits of a word (the tag takes the other 31 - ib)
};

// Sample queries per frame and how many of them must have a bitwise copy in the tree for the frame to take the pass in the
// automatic mode.  Matcher stage per 200 x 50k frames by share of queries WITHOUT a copy: 0 %: 0.52 ms, 1 %:
// 0.72, 5 %: 0.77, 25 %: 1.26, 100 %: 1.21 with "any of 8 samples" (round 4) -- the search alone: 1.10 at every share, so the
// pass loses 0.1-0.2 ms from ~10 % on.  Round 5 tried the obvious repair: 16 samples, at least 15 found -- taken
// with probability 0.99 / 0.81 / 0.06 / 0 at 1 / 5 / 25 / 100 % -- and measured it WORSE at every share: 0.77 / 1.10 / 1.32 /
// 1.29 ms (frames of one call split over two routes keep both chains of kernels busy, and sixteen scalar sample loads per
// wave lengthen the row-hash kernel by 0.08 ms).  So the rule stays "any of 8"; vo_match_set_mode(3) is the choice for data
// known to carry recomputed descriptors.  The plumbing for a threshold stays (HashArgs::sample_min).
constexpr int HJ_SAMPLES = 8, HJ_SAMPLE_MIN = 1;
__device__ __forceinline__ bool hash_frame_takes_pass(const HashArgs& a, int f) { return __popc(a.sample_mask[f]) >= a.sample_min; }
__device__ __forceinline__ bool row_safe(const Row10& r) {       // every |x| in [2^-40, inf): an unsigned range test on the bits
  unsigned worst = 0;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const unsigned t = (__float_as_uint(r.v[k]) & 0x7fffffffu) - 0x2b800000u;       // 2^-40 = 0x2b800000; smaller magnitudes wrap around
    worst = t > worst ? t : worst;
  }
  return worst < 0x7f800000u - 0x2b800000u;                     // below +inf (NaN payloads lie above)
}
__device__ __forceinline__ bool rows_equal(const Row10& a, const Row10& b) {
  unsigned d = 0;
#pragma unroll
  for (int k = 0; k < 10; ++k) d |= __float_as_uint(a.v[k]) ^ __float_as_uint(b.v[k]);
  return d == 0u;                                  // bitwise: for safe rows (no zeros, no NaN) the same as float equality
}
__device__ __forceinline__ unsigned rotl32(unsigned x, int r) { return (x << r) | (x >> (32 - r)); }
// One 32-bit hash per row; its top bits pick the part, its low bits the home slot, the bits in between are the tag.
// Rotate-and-combine over the ten words (alternating add and xor, two full-rate instructions per word) and two
// multiply-xorshift rounds at the end -- 32-bit multiplies are quarter-rate: MurmurHash3's thirty of them made an earlier
// form of the build, in which every part's workgroup hashed every row, VALU-bound at 65 us per workgroup and 50k rows.
// Nothing but speed depends on the hash's quality: a part that overflows is stored empty, its queries go to the search.
// (tests/test_gpu_hashfirst.py restates it in numpy to build trees that overflow a part on purpose.)
__device__ __forceinline__ unsigned row_hash(const Row10& r) {
  unsigned x = __float_as_uint(r.v[0]);
#pragma unroll
  for (int k = 1; k < 10; ++k) {
    const unsigned w = __float_as_uint(r.v[k]);
    x = (k & 1) ? rotl32(x, 7) + w : rotl32(x, 11) ^ w;
  }
  x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
  return x;
}
__device__ __forceinline__ unsigned hj_part(unsigned h, int log2p) { return log2p ? h >> (32 - log2p) : 0u; }
// home slot (even: the lookup's 16-byte window from it is 8-byte aligned) and tag of a hash
template <int LOG2S> __device__ __forceinline__ unsigned hj_home(unsigned h) { return h & ((1u << LOG2S) - 2u); }
template <int LOG2S> __device__ __forceinline__ unsigned hj_tag(unsigned h, int ib) { return (h >> LOG2S) & ((1u << (31 - ib)) - 1u); }

// Streaming 40-byte rows: a lane that loads "its" row issues three requests (16 + 16 + 8 bytes) which the texture path
// takes one lane at a time (~1 lane-request per clock and CU: 150 k requests = 62 us for one workgroup's pass over 50k rows,
// measured).  So a wave loads 64 consecutive rows as ONE contiguous 2560-byte run -- five 8-byte loads per lane, lane l
// taking bytes 512 k + 8 l -- and transposes through a 2560-byte LDS strip of its own: row r of the run is float2 elements
// 5 r .. 5 r + 4 (a 40-byte stride over 8-byte reads: conflict-free per half-wave).
// A row fetched inside the build's insertion loop (a tag that agrees: rare).  Written as one block with its own wait, so
// that the compiler's wait-count bookkeeping never sees a load of unknown standing inside the streaming loop -- with a plain
// load there it waits for vmcnt(0), i.e. for all the runs requested ahead, at the top of every pass.
__device__ __forceinline__ Row10 load_row_now(const float* p) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  f4v a, b; f2v c;
  asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\t"
               "global_load_dwordx2 %2, %3, off offset:32\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p) : "memory");
  Row10 r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  r.v[8] = c.x; r.v[9] = c.y;
  return r;
}
struct RowRun { float2 v[5]; };
__device__ __forceinline__ RowRun run_load(const float* set, int n_rows, int chunk) {       // rows 64 chunk .. 64 chunk + 63 of the set (n_rows > 0)
  RowRun r;
  const int lane = threadIdx.x & 63;
  const float2* base = reinterpret_cast<const float2*>(set);
  const long long limit = 5ll * n_rows;                        // float2 elements of the set
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    // UNCONDITIONAL loads (an element beyond the set reads element 0 instead; the caller ignores rows >= n_rows): with
    // predicated loads the compiler cannot count what is in flight and waits for everything at every use
    const long long e = (long long)chunk * HJ_STRIP + 64 * k + lane;
    r.v[k] = base[e < limit ? e : 0];
  }
  return r;
}
__device__ __forceinline__ Row10 run_row(const RowRun& r, float2* strip) {                   // this lane's row of the run
  // One wave, one strip: the LDS executes a wave's instructions in order, so the reads below see the writes above them and
  // the next run's writes cannot overtake these reads; the compiler keeps the order because the addresses may alias.  (No
  // fence: a release fence here waits for vmcnt(0), i.e. for the runs that were requested ahead -- 96 instead of 68 us.)
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < 5; ++k) strip[64 * k + lane] = r.v[k];
  __builtin_amdgcn_wave_barrier();
  Row10 o;
#pragma unroll
  for (int k = 0; k < 5; ++k) { const float2 t = strip[5 * lane + k]; o.v[2 * k] = t.x; o.v[2 * k + 1] = t.y; }
  return o;
}

// step 1: the hash of every tree row -- a plain map over the rows, the whole chip streams them once -- written as (hash, index)
// records into one QUEUE PER PART of the frame's table: a workgroup ranks its 256 rows inside their parts with LDS atomics,
// reserves a run in every part's queue with one global atomic per part, and stores its records there.  (The order of a
// queue's records varies from run to run; nothing downstream depends on it: the table keeps the lowest index of equal rows
// whatever the order of insertion.)  A queue holds 1.5 x the average part; the cursor counts beyond it, which step 2 reads
// as "this part overflowed".
constexpr int HJ_MAXP = 32;                         // parts per frame, at most (hash_plan: log2p <= 5)
__global__ __launch_bounds__(256) void hash_rows_kernel(HashArgs a) {
  int f, blk;
  if (!xcd_frame_block(a.hblocks, a.n_frames, f, blk)) return;
  const float* tree; const float* qry; int nt, nq;
  cell_sets(a, f, tree, qry, nt, nq);
  if (blk * 256 >= nt) return;
  __shared__ float2 s_strip[4][HJ_STRIP];
  __shared__ int s_cnt[HJ_MAXP], s_base[HJ_MAXP];
  const int tid = threadIdx.x;
  if (tid < HJ_MAXP) s_cnt[tid] = 0;
  const int i = blk * 256 + tid;
  const Row10 r = run_row(run_load(tree, nt, i >> 6), s_strip[tid >> 6]);
  const unsigned h = row_hash(r);
  const unsigned part = hj_part(h, a.log2p);
  // Does this data copy appearances at all?  Eight queries of the frame, evenly spread, are looked for while the tree streams
  // past anyway (first component against eight scalars; a row only when that agrees): descriptors recomputed per frame have
  // no copies, and a frame in which none of the eight is found leaves the tables and the lookup out (hash_table_kernel,
  // hash_probe_kernel) -- 1.41 -> 1.2 ms for the matcher stage of 200 x 50k such frames, 1.10 without the pass.  A wrong guess
  // costs time only: whatever the pass leaves open, the search finds.
  if (nq > 0) {
    const int step = nq / HJ_SAMPLES;
    unsigned hit = 0;
#pragma unroll
    for (int k = 0; k < HJ_SAMPLES; ++k) hit |= (__float_as_uint(qry[10 * (size_t)(k * step)]) == __float_as_uint(r.v[0]) ? 1u : 0u) << k;
    if (i < nt && hit) {
      for (int k = 0; k < HJ_SAMPLES; ++k)
        if ((hit >> k) & 1u) {
          const Row10 sq = load_row_now(qry + 10 * (size_t)(k * step));
          if (rows_equal(sq, r)) atomicOr(&a.sample_mask[f], 1u << k);
        }
    }
  }
  __syncthreads();
  int pos = 0;
  if (i < nt) pos = atomicAdd(&s_cnt[part], 1);
  __syncthreads();
  if (tid < (1 << a.log2p)) {
    const int c = s_cnt[tid];
    s_base[tid] = c ? atomicAdd(&a.cursors[(size_t)f * HJ_MAXP + tid], c) : 0;
  }
  __syncthreads();
  if (i < nt) {
    const int slot = s_base[part] + pos;
    if (slot < a.qcap) a.queues[((size_t)f << a.log2p) * a.qcap + (size_t)part * a.qcap + slot] = make_uint2(h, (unsigned)i);
  }
}

// step 2: one workgroup per (frame, part) takes its queue -- every lane a record -- and inserts into an open-addressing table in
// LDS, then stores the part as one contiguous image.  (Earlier forms, same box, 200 x 50k: every part's workgroup streaming all
// ROWS and hashing them itself, 287-317 us -- quarter-full compare-and-swap chains; all HASHES read by every part's workgroup and
// compacted by ballot into a per-wave queue, 85 us; the queues written by step 1: ~35 us.)  A part whose queue overflowed, or
// whose table cannot take a row within HJ_PROBES slots (a degenerate hash), is stored EMPTY: all its queries stay open and go
// to the search -- a table that lacks some rows could hold a copy's higher index without the lower one.
constexpr unsigned HJ_PROBES = 512;
template <int LOG2S>
__global__ __launch_bounds__(HJ_THREADS) void hash_table_kernel(HashArgs a) {
  constexpr unsigned S = 1u << LOG2S;
  int f, part;
  if (!xcd_frame_block(1 << a.log2p, a.n_frames, f, part)) return;
  if (!hash_frame_takes_pass(a, f)) return;                // too few copies in this frame's data (hash_rows_kernel): no table, no lookup
  const float* tree; const float* qry; int nt, nq;
  cell_sets(a, f, tree, qry, nt, nq);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __shared__ __attribute__((aligned(16))) unsigned s_tab[S];
  __shared__ int s_ovf;
  if (tid == 0) s_ovf = 0;
  for (unsigned k = tid; k < S / 4; k += HJ_THREADS) reinterpret_cast<uint4*>(s_tab)[k] = make_uint4(HJ_EMPTY, HJ_EMPTY, HJ_EMPTY, HJ_EMPTY);
  const int filled = a.cursors[(size_t)f * HJ_MAXP + part];
  const int count = filled < a.qcap ? filled : a.qcap;
  __syncthreads();
  const unsigned imask = (1u << a.ib) - 1u;
  const uint2* q = a.queues + ((size_t)f << a.log2p) * a.qcap + (size_t)part * a.qcap;
  if (filled <= a.qcap) {
    const int n_groups = (count + 63) >> 6;
    uint2 e = make_uint2(0u, 0u);
    if (wave < n_groups) e = q[64 * wave + lane < count ? 64 * wave + lane : 0];
    for (int g = wave; g < n_groups; g += HJ_NW) {
      const uint2 cur = e;
      const int nxt = 64 * (g + HJ_NW) + lane;
      e = q[nxt < count ? nxt : 0];                          // (unconditional: the next group's record, or record 0 again)
      if (64 * g + lane < count) {
        const unsigned h = cur.x;
        const unsigned word = (hj_tag<LOG2S>(h, a.ib) << a.ib) | cur.y;
        unsigned s = hj_home<LOG2S>(h);
        bool done = false;
        for (unsigned n = 0; n < HJ_PROBES; ++n) {
          const unsigned old = atomicCAS(&s_tab[s], HJ_EMPTY, word);
          if (old == HJ_EMPTY) { done = true; break; }
          if ((old >> a.ib) == (word >> a.ib)) {
            // same tag: the same row?  (the entry's index may be lowered meanwhile -- by a copy of the same row)
            const Row10 o = load_row_now(tree + 10 * (size_t)(old & imask)), r = load_row_now(tree + 10 * (size_t)(word & imask));
            if (rows_equal(o, r)) { atomicMin(&s_tab[s], word); done = true; break; }
          }
          s = (s + 1u) & (S - 1u);
        }
        if (!done) s_ovf = 1;
      }
    }
  }
  __syncthreads();
  const bool drop = filled > a.qcap || s_ovf != 0;
  uint4* dst = reinterpret_cast<uint4*>(a.tables + f * a.tables_stride + ((size_t)part << LOG2S));
  const uint4 none = make_uint4(HJ_EMPTY, HJ_EMPTY, HJ_EMPTY, HJ_EMPTY);
  // (two loops: a select between the constant and the LDS word made the compiler park the constant in scratch memory and fetch
  // either through a flat load, waiting for it in every pass)
  if (drop) { for (unsigned k = tid; k < S / 4; k += HJ_THREADS) dst[k] = none; }
  else { for (unsigned k = tid; k < S / 4; k += HJ_THREADS) dst[k] = reinterpret_cast<const uint4*>(s_tab)[k]; }
}

// step 3: the lookup, in query order.  A query's chain is three dependent round trips -- its row, its table word, the
// candidate's row.  A lane can carry HJ_Q queries whose round trips fly together; measured at 200 x 50k frames: 277 us
// with one query per lane, 297 with two, 360 with four (fewer waves) -- the kernel is not short of requests in flight, it
// runs at the rate the XCD's L2 answers scattered line requests (2.8 per query: 0.3 for its row, 1 table word, 1.5 for the
// candidate's 40-byte row; 1.0 GB of HBM traffic per call, 3.6 TB/s).  So one query per lane.
// Fast path: the home slot and the three behind it (one 16-byte load) either end the chain (an empty slot) or name a candidate
// whose row settles it; everything else -- a longer chain, a tag that agrees on a different row -- walks the chain in full.
#ifndef VO_HJ_Q
#define VO_HJ_Q 1
#endif
constexpr int HJ_Q = VO_HJ_Q;
template <int LOG2S>
__global__ __launch_bounds__(256) void hash_probe_kernel(HashArgs a) {
  constexpr unsigned S = 1u << LOG2S, MASK = S - 1u;
  int f, blk;
  if (!xcd_frame_block(a.qblocks, a.n_frames, f, blk)) return;
  const float* tree; const float* qry; int nt, nq;
  cell_sets(a, f, tree, qry, nt, nq);
  if (blk * (256 * HJ_Q) >= nq) return;                   // (uniform)
  if (!hash_frame_takes_pass(a, f)) {                     // (uniform) a frame without (enough) copies: every query stays open
    unsigned long long* best = a.best + f * a.best_stride;
    int* seg = reinterpret_cast<int*>(a.queues + f * a.queue_stride);
    const int base = blk * (256 * HJ_Q), live = nq - base < 256 * HJ_Q ? nq - base : 256 * HJ_Q;
    for (int t = threadIdx.x; t < live; t += 256) {
      best[base + t] = ((unsigned long long)__float_as_uint(a.r2) << 32) | 0xffffffffull;
      seg[base + t] = base + t;
    }
    if (threadIdx.x == 0) seg[a.seg_counts + blk] = live;
    return;
  }
  __shared__ int s_open;
  __shared__ float2 s_strip[4][HJ_STRIP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) s_open = 0;
  const unsigned imask = (1u << a.ib) - 1u;
  const unsigned* tab = a.tables + f * a.tables_stride;
  // query u of this lane: chunk (4 blk + wave) HJ_Q + u, i.e. the wave's HJ_Q consecutive runs of 64 queries
  int j[HJ_Q];
  RowRun run[HJ_Q];
#pragma unroll
  for (int u = 0; u < HJ_Q; ++u) {
    const int chunk = (4 * blk + wave) * HJ_Q + u;
    j[u] = 64 * chunk + lane;
    run[u] = run_load(qry, nq, chunk);
  }
  Row10 q[HJ_Q];
  unsigned h[HJ_Q], found[HJ_Q];
  bool look[HJ_Q];
  uint4 w03[HJ_Q];
#pragma unroll
  for (int u = 0; u < HJ_Q; ++u) {
    q[u] = run_row(run[u], s_strip[wave]);
    found[u] = 0xffffffffu;
    look[u] = j[u] < nq && nt > 0 && a.r2 > 0.f && row_safe(q[u]);
    h[u] = row_hash(q[u]);
  }
#pragma unroll
  for (int u = 0; u < HJ_Q; ++u) {                         // the table words: unconditional loads (a lane that does not look reads slot 0)
    const unsigned* t = tab + ((size_t)hj_part(h[u], a.log2p) << LOG2S);
    w03[u] = *reinterpret_cast<const uint4*>(t + (look[u] ? hj_home<LOG2S>(h[u]) : 0u));   // (8-byte aligned: the home slot is even)
  }
  unsigned cand[HJ_Q];
  bool slow[HJ_Q];
#pragma unroll
  for (int u = 0; u < HJ_Q; ++u) {
    const unsigned tag = hj_tag<LOG2S>(h[u], a.ib);
    // the chain from the home slot, four slots of it: it ends at the first empty slot; the first tag that agrees before that
    // names the candidate.  Neither within the window -> walk.  (A home slot two before the part's end: slots 2 and 3 of the
    // load lie behind the part -- the chain wraps --, so the window is two slots there.)
    const bool four = hj_home<LOG2S>(h[u]) != S - 2u;
    const unsigned ws[4] = {w03[u].x, w03[u].y, w03[u].z, w03[u].w};
    bool open_end = true;                                  // no end, no candidate so far
    cand[u] = 0xffffffffu;
    bool ended = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool seen = k < 2 || four;
      const bool e = seen && ws[k] == HJ_EMPTY, m = seen && ws[k] != HJ_EMPTY && (ws[k] >> a.ib) == tag;
      if (open_end && e) { ended = true; open_end = false; }
      if (open_end && m) { cand[u] = ws[k] & imask; open_end = false; }
    }
    // settled without a candidate: the chain ends inside the window.  Not settled: no candidate and no end -> walk.
    slow[u] = look[u] && cand[u] == 0xffffffffu && !ended;
    if (!look[u]) cand[u] = 0xffffffffu;
  }
  Row10 t[HJ_Q];
#pragma unroll
  for (int u = 0; u < HJ_Q; ++u) t[u] = load_row(tree + 10 * (size_t)(cand[u] != 0xffffffffu ? cand[u] : 0u));   // unconditional
#pragma unroll
  for (int u = 0; u < HJ_Q; ++u) {
    if (cand[u] != 0xffffffffu) {
      if (rows_equal(t[u], q[u])) found[u] = cand[u];      // one entry per distinct row: the chain holds no other copy
      else slow[u] = true;                                 // the tag of another row: walk
    }
  }
#pragma unroll
  for (int u = 0; u < HJ_Q; ++u) {
    if (slow[u]) {                                         // rare: the chain from the home slot to the first empty one
      const unsigned tag = hj_tag<LOG2S>(h[u], a.ib);
      const unsigned* tp = tab + ((size_t)hj_part(h[u], a.log2p) << LOG2S);
      unsigned s = hj_home<LOG2S>(h[u]);
      for (unsigned n = 0; n < S; ++n) {
        const unsigned w = tp[s];
        if (w == HJ_EMPTY) break;
        if ((w >> a.ib) == tag) {
          const unsigned c = w & imask;
          if (rows_equal(load_row(tree + 10 * (size_t)c), q[u])) { found[u] = c; break; }
        }
        s = (s + 1u) & MASK;
      }
    }
  }
  __syncthreads();
  unsigned long long* best = a.best + f * a.best_stride;
  const bool tree_is_1 = a.d_n1 ? tree == a.tree + f * a.tree_stride : a.tree_is_1 != 0;      // (cell_sets swaps the sets when set 2 is the tree)
  // The frame's open queries are counted and listed WITHOUT a global atomic: a workgroup ranks its own in LDS, stores their
  // indices in its own segment (the part queues of step 1 are free by now) and its count beside them; hash_open_kernel adds
  // the counts up per frame.  (One atomicAdd per workgroup on the frame's counter -- 196 on one address per frame, executed
  // at the memory side -- cost the lookup 60 us per 200 x 50k frames as soon as most workgroups held an open query.)
  int* seg = reinterpret_cast<int*>(a.queues + f * a.queue_stride);
#pragma unroll
  for (int u = 0; u < HJ_Q; ++u) {
    const bool live = j[u] < nq;
    const bool open = live && found[u] == 0xffffffffu;
    const unsigned long long bal = __ballot(open);
    const int n_w = __popcll(bal);
    int wbase = 0;
    if (lane == 0 && n_w > 0) wbase = atomicAdd(&s_open, n_w);
    const int slot = __shfl(wbase, 0) + __popcll(bal & ((1ull << lane) - 1ull));
    if (open) seg[blk * (256 * HJ_Q) + slot] = j[u];
    if (live) best[j[u]] = open ? (((unsigned long long)__float_as_uint(a.r2) << 32) | 0xffffffffull) : (unsigned long long)found[u];
    // pair j at slot j: where it belongs when no query of the frame stays open (match_count / match_scatter then skip the frame)
    if (live && !open && a.out_pairs)
      reinterpret_cast<int2*>(a.out_pairs + 2 * f * a.out_stride)[j[u]] = tree_is_1 ? make_int2((int)found[u], j[u]) : make_int2(j[u], (int)found[u]);
  }
  __syncthreads();
  if (tid == 0) seg[a.seg_counts + blk] = s_open;
}

// step 4: one workgroup per frame adds the lookup workgroups' counts up (unres[f]) and, when there are few enough, strings
// their segments together into the frame's list of open queries (open_collect_kernel)
__global__ __launch_bounds__(256) void hash_open_kernel(HashArgs a) {
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* tree; const float* qry; int nt, nq;
  cell_sets(a, f, tree, qry, nt, nq);
  const int nblk = (nq + 256 * HJ_Q - 1) / (256 * HJ_Q);
  const int* seg = reinterpret_cast<const int*>(a.queues + f * a.queue_stride);
  const int* cnt = seg + a.seg_counts;
  __shared__ int s_w[4];
  __shared__ int s_carry;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  // pass 1: the total
  int mine = 0;
  for (int b = tid; b < nblk; b += 256) mine += cnt[b];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
  if (lane == 0) s_w[wave] = mine;
  __syncthreads();
  const int total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  if (tid == 0) a.unres[f] = total;
  if (total == 0 || total > OPEN_MAX) return;              // (uniform) no list wanted
  __syncthreads();
  // pass 2: exclusive prefix of the counts, 256 workgroups per trip, and the segments copied behind one another
  int* list = a.open_list + (size_t)f * OPEN_MAX;
  for (int b0 = 0; b0 < nblk; b0 += 256) {
    const int b = b0 + tid;
    const int c = b < nblk ? cnt[b] : 0;
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int at = s_carry + incl - c;
#pragma unroll
    for (int w = 0; w < 4; ++w) at += w < wave ? s_w[w] : 0;
    for (int k = 0; k < c; ++k) list[at + k] = seg[b * (256 * HJ_Q) + k];      // (few: <= OPEN_MAX over the whole frame)
    __syncthreads();
    if (tid == 0) s_carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
}

// ---- few open queries: the tree streamed past them -------------------------------------------------------------------
// A frame whose exact-duplicate pass left only a FEW queries open (new landmarks: a per cent or two of a tracking frame) would
// still pay the whole sorted search -- its cost is sorting the TREE into cells, whatever the number of queries (measured,
// 200 x 50k frames: 1 % open queries 1.11 ms for the matcher stage against 0.53 with none and 1.12 without the pass).  For such
// a frame the roles are turned round: the open queries (<= OPEN_MAX, and <= nq / open_div) are ordered by the coarse bin
// (c0, c1) of the SAME grid as the cell-hash search (one workgroup per frame, open_collect_kernel), and the tree is streamed
// ONCE past them (open_scan_kernel): a tree point t can only be within the radius of queries whose cells lie in
// [cell(t - R), cell(t + R)] per component -- fl(t - R) <= q <= fl(t + R) for every float q with |t - q| < r < R because
// rounding is monotone, and the cell function is monotone -- i.e. of <= 3 runs of the ordered list (contiguous along c1).  Every
// candidate goes through the search's filter (one v_sad_u8 on the quantised components 0..3, proof at the cell variant) and a
// survivor is decided from the two full rows in the reference's operation order; the result enters the query's key by
// atomicMin on (distance bits, tree index) -- the smallest distance, the lowest index among equals, strictly inside the radius:
// the reference's rule, whatever the order in which tree points arrive.  ~9 M / 400 candidates per tree point (M open queries):
// at M = 500 a pass over the rows (~0.1 ms per 200 x 50k frames) instead of 0.6 ms of sorting.
// A frame with more open queries, or whose open queries crowd into few bins (the pass would degenerate into M x nt filter
// tests), keeps the sorted search: todo[f] = unres[f].
constexpr int OPEN_THREADS = 1024;
constexpr int OPEN_PER_THREAD = (OPEN_MAX + OPEN_THREADS - 1) / OPEN_THREADS;
constexpr int OPEN_BINS = HNC * HNC * HNC;            // (c0, c1, c2), numbered (c0 * HNC + c1) * HNC + c2 whatever the grid
constexpr int OPEN_BPT = 8;                           // bins per thread in the scan of their counts
constexpr int OPEN_SROW = OPEN_THREADS * OPEN_BPT + 8;   // u16 entries of the start table (bins beyond the grid hold the total)
static_assert(OPEN_BINS <= OPEN_THREADS * OPEN_BPT && OPEN_RECS < 65536, "bins per thread / 16-bit starts");
static_assert(OPEN_SROW_WS == OPEN_SROW, "workspace block of the start table");
// one workgroup per frame: the open queries, each entered into the <= 9 columns (c0, c1) of ITS box at its own c2 -- a tree point
// then finds every query it can be within the radius of in ONE run of the list, its own column along c2 (open_scan_kernel)
__global__ __launch_bounds__(OPEN_THREADS) void open_collect_kernel(CellArgs a) {
  const int f = blockIdx.x;
  const int open = a.unres[f];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (open == 0) { if (tid == 0) a.todo[f] = 0; return; }
  const float* tree; const float* qry; int nt, nq;
  cell_sets(a, f, tree, qry, nt, nq);
  int cap = a.open_div > 0 ? nq / a.open_div : 0;
  cap = cap < OPEN_MAX ? cap : OPEN_MAX;
  if (open > cap || nt <= 0) { if (tid == 0) a.todo[f] = open; return; }       // (uniform)
  char* ws = a.ws + f * a.ws_stride;
  const int* list = a.open_list + (size_t)f * OPEN_MAX;  // (all `open` entries are there: open <= OPEN_MAX)
  __shared__ int s_cnt[OPEN_THREADS * OPEN_BPT];
  __shared__ unsigned s_out[OPEN_RECS];
  __shared__ int s_max;
  __shared__ int s_w[OPEN_THREADS / 64];
#pragma unroll
  for (int k = 0; k < OPEN_BPT; ++k) s_cnt[tid + k * OPEN_THREADS] = 0;
  if (tid == 0) s_max = 0;
  const int m = open;
  __syncthreads();
  const CellParams cp = *reinterpret_cast<const CellParams*>(ws + a.w.cp);
  int qi[OPEN_PER_THREAD], c2[OPEN_PER_THREAD], c0lo[OPEN_PER_THREAD], c1lo[OPEN_PER_THREAD], w0[OPEN_PER_THREAD], w1[OPEN_PER_THREAD];
  int rank[OPEN_PER_THREAD][9];
  unsigned word[OPEN_PER_THREAD];
  // (three phases, each over all of the thread's queries, so that their round trips overlap: indices, rows, then cells and ranks)
#pragma unroll
  for (int u = 0; u < OPEN_PER_THREAD; ++u) {
    const int k = tid + u * OPEN_THREADS;
    qi[u] = list[k < m ? k : 0];
    qi[u] = (unsigned)qi[u] < (unsigned)nq ? qi[u] : 0;    // (always inside: the lookup wrote it; never an address from an unchecked word)
  }
  float v10[OPEN_PER_THREAD][10];
#pragma unroll
  for (int u = 0; u < OPEN_PER_THREAD; ++u) load10(qry + 10 * (size_t)qi[u], v10[u]);
#pragma unroll
  for (int u = 0; u < OPEN_PER_THREAD; ++u) {
    const int k = tid + u * OPEN_THREADS;
    c2[u] = 0; c0lo[u] = 0; c1lo[u] = 0; w0[u] = -1; w1[u] = -1; word[u] = 0;      // (w0 < 0: no query here)
    if (k < m) {
      const float x0 = pick10(v10[u], cp.dim[0]), x1 = pick10(v10[u], cp.dim[1]);
      c0lo[u] = cell_of(x0 - cp.R, cp.lo[0], cp.scale[0], cp.nc[0]);
      c1lo[u] = cell_of(x1 - cp.R, cp.lo[1], cp.scale[1], cp.nc[1]);
      w0[u] = cell_of(x0 + cp.R, cp.lo[0], cp.scale[0], cp.nc[0]) - c0lo[u];
      w1[u] = cell_of(x1 + cp.R, cp.lo[1], cp.scale[1], cp.nc[1]) - c1lo[u];
      c2[u] = cell_of(pick10(v10[u], cp.dim[2]), cp.lo[2], cp.scale[2], cp.nc[2]);
      word[u] = filter_word(v10[u][0], v10[u][1], v10[u][2], v10[u][3], cp);
    }
#pragma unroll
    for (int i0 = 0; i0 < 3; ++i0)
#pragma unroll
      for (int i1 = 0; i1 < 3; ++i1) {
        rank[u][3 * i0 + i1] = -1;
        if (i0 <= w0[u] && i1 <= w1[u])
          rank[u][3 * i0 + i1] = atomicAdd(&s_cnt[((c0lo[u] + i0) * HNC + c1lo[u] + i1) * HNC + c2[u]], 1);
      }
  }
  __syncthreads();
  // exclusive scan of the bins' counts (OPEN_BPT consecutive bins per thread), the fullest bin, and whether a box was wider
  // than three cells in a component (cells are >= R wide, so it cannot be -- but nothing may hang on that: then the sorted search)
  int c[OPEN_BPT], sum = 0, most = 0;
#pragma unroll
  for (int k = 0; k < OPEN_BPT; ++k) { c[k] = s_cnt[tid * OPEN_BPT + k]; sum += c[k]; most = c[k] > most ? c[k] : most; }
#pragma unroll
  for (int u = 0; u < OPEN_PER_THREAD; ++u) if (w0[u] > 2 || w1[u] > 2) most = 0x7fffffff;
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  if (lane == 63) s_w[wave] = incl;
  if (most > 0) atomicMax(&s_max, most);
  __syncthreads();
  // crowded bins: the stream would test every tree point near them against all their queries
  if (s_max > 32 + m / 16) { if (tid == 0) a.todo[f] = open; return; }         // (uniform)
  int ex = incl - sum, total = 0;
#pragma unroll
  for (int w = 0; w < OPEN_THREADS / 64; ++w) { ex += w < wave ? s_w[w] : 0; total += s_w[w]; }
  unsigned short* ostart = reinterpret_cast<unsigned short*>(ws + a.w.open_start);
  {
    unsigned pk[OPEN_BPT / 2];
    int run = ex;
#pragma unroll
    for (int k = 0; k < OPEN_BPT; ++k) {
      s_cnt[tid * OPEN_BPT + k] = run;                    // (this thread's own counters, read above)
      if (k & 1) pk[k / 2] |= (unsigned)run << 16; else pk[k / 2] = (unsigned)run;
      run += c[k];
    }
    reinterpret_cast<uint4*>(ostart)[tid] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  }
  if (tid < 8) ostart[OPEN_THREADS * OPEN_BPT + tid] = (unsigned short)total;   // (the entries behind the last bin: the list's length)
  if (tid == 0) a.todo[f] = 0;
  __syncthreads();
  // the records in list order, through LDS: placed there by bin, then copied out as one contiguous run -- first the filter
  // words, then the indices.  (Stored straight to memory they are 2 x 9 m scattered 4-byte stores from ONE workgroup: 84 of
  // this kernel's 108 us per 200 frames with 2500 open queries each.)
  unsigned* oword = reinterpret_cast<unsigned*>(ws + a.w.open_rec);
  unsigned* oidx = oword + OPEN_RECS;
  int slot[OPEN_PER_THREAD][9];
#pragma unroll
  for (int u = 0; u < OPEN_PER_THREAD; ++u)
#pragma unroll
    for (int i0 = 0; i0 < 3; ++i0)
#pragma unroll
      for (int i1 = 0; i1 < 3; ++i1) {
        const int r = rank[u][3 * i0 + i1];
        slot[u][3 * i0 + i1] = r >= 0 ? s_cnt[((c0lo[u] + i0) * HNC + c1lo[u] + i1) * HNC + c2[u]] + r : -1;
      }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();                                      // (pass 0: the starts are read; pass 1: the words are copied out)
#pragma unroll
    for (int u = 0; u < OPEN_PER_THREAD; ++u)
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (slot[u][k] >= 0) s_out[slot[u][k]] = pass == 0 ? word[u] : (unsigned)qi[u];
    __syncthreads();
    unsigned* dst = pass == 0 ? oword : oidx;
    for (int k = tid; k < total; k += OPEN_THREADS) dst[k] = s_out[k];
  }
}

// The tree streamed past the list.  A workgroup stages the frame's start table and filter words ONCE (72 KB at most: one
// workgroup per CU, so nothing hides that round trip but the length of the work behind it) and every wave then takes
// `runs_per_wave` consecutive runs of 64 rows, the next run's loads in flight while the current one is looked up.
constexpr int OPEN_SCAN_THREADS = 1024;
constexpr int OPEN_SCAN_WAVES = OPEN_SCAN_THREADS / 64;
__global__ __launch_bounds__(OPEN_SCAN_THREADS) void open_scan_kernel(CellArgs a, int tblocks, int runs_per_wave) {
  int f, blk;
  if (!xcd_frame_block(tblocks, a.n_frames, f, blk)) return;
  const int m = a.unres[f];
  if (m == 0 || a.todo[f] != 0) return;                   // no open query, or the sorted search takes them
  const float* tree; const float* qry; int nt, nq;
  cell_sets(a, f, tree, qry, nt, nq);
  const int rows_per_wg = OPEN_SCAN_WAVES * 64 * runs_per_wave;
  if ((long long)blk * rows_per_wg >= nt) return;         // (uniform)
  char* ws = a.ws + f * a.ws_stride;
  __shared__ __attribute__((aligned(16))) unsigned short s_start[OPEN_SROW];
  __shared__ unsigned s_word[OPEN_RECS + 4];
  __shared__ float2 s_strip[OPEN_SCAN_WAVES][HJ_STRIP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int chunk0 = (blk * OPEN_SCAN_WAVES + wave) * runs_per_wave;
  const int n_chunks = (nt + 63) >> 6;
  RowRun cur = run_load(tree, nt, chunk0 < n_chunks ? chunk0 : 0);
  RowRun cur2 = run_load(tree, nt, chunk0 + 1 < n_chunks ? chunk0 + 1 : 0);
  const unsigned short* ostart = reinterpret_cast<const unsigned short*>(ws + a.w.open_start);
  const unsigned* oword = reinterpret_cast<const unsigned*>(ws + a.w.open_rec);
  const unsigned* oidx = oword + OPEN_RECS;
  for (int k = tid; k < OPEN_SROW / 8; k += OPEN_SCAN_THREADS) reinterpret_cast<uint4*>(s_start)[k] = reinterpret_cast<const uint4*>(ostart)[k];
  const int n_stage = 9 * m < OPEN_RECS ? 9 * m : OPEN_RECS;      // (>= the list's length, known without waiting for it)
  for (int k = tid; k < n_stage; k += OPEN_SCAN_THREADS) s_word[k] = oword[k];
  const CellParams cp = *reinterpret_cast<const CellParams*>(ws + a.w.cp);
  unsigned long long* best = a.best + f * a.best_stride;
  const int T = cp.T;
  __syncthreads();
  // Two runs per pass: the kernel is a chain of LDS round trips (strip, start table, filter words), and two independent
  // chains in a wave hide each other's.  The pair after this one is in flight meanwhile.
  auto lookup = [&](const float* mine, int& b, int& e) {       // this lane's run of the list: [b, e)
    // the hashed components straight from the strip (this wave's, still holding the run): one LDS read each instead of a
    // ten-way select on registers
    const float x0 = mine[cp.dim[0]], x1 = mine[cp.dim[1]], x2 = mine[cp.dim[2]];
    const int c0 = cell_of(x0, cp.lo[0], cp.scale[0], cp.nc[0]), c1 = cell_of(x1, cp.lo[1], cp.scale[1], cp.nc[1]);
    const int c2lo = cell_of(x2 - cp.R, cp.lo[2], cp.scale[2], cp.nc[2]), c2hi = cell_of(x2 + cp.R, cp.lo[2], cp.scale[2], cp.nc[2]);
    const int base = (c0 * HNC + c1) * HNC;
    b = s_start[base + c2lo];                             // (both unconditional: one round trip)
    e = s_start[base + c2hi + 1];
  };
  // the decision: the reference's unfused left-to-right sum (brute_force_search.h:34), strictly inside the radius; the
  // smallest (distance, tree index) wins whatever the order in which tree points arrive
  auto decide = [&](const Row10& t, const int ti, const int p) {
    const unsigned qi = oidx[p];
    const Row10 q = load_row(qry + 10 * (size_t)qi);
    float d = t.v[0] - q.v[0];
    float s2 = d * d;
#pragma unroll
    for (int c = 1; c < 10; ++c) { d = t.v[c] - q.v[c]; s2 += d * d; }
    if (s2 < a.r2)
      atomicMin(&best[qi], ((unsigned long long)__float_as_uint(s2) << 32) | (unsigned long long)(unsigned)ti);
  };
  // A filter survivor costs two dependent memory round trips (the query's index, then its row) -- and with a per cent of the
  // queries open most runs of 64 tree rows hold one: decided on the spot they were the kernel's time (16 waves per CU cannot
  // hide that many).  So a lane PARKS its first two survivors (tree index, list position) and decides them after its last
  // run, all lanes' loads in flight together; the tree row is fetched again then (L2: the frame was just streamed).
  int park_ti0 = -1, park_p0 = 0, park_ti1 = -1, park_p1 = 0;
  auto visit = [&](const Row10& t, const int ti, int b, const int e) {
    if (b >= e) return;
    const unsigned tw = filter_word(t.v[0], t.v[1], t.v[2], t.v[3], cp);
    for (; b < e; b += 4) {
      unsigned w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = s_word[b + k];    // (s_word is padded: a read behind the list is ignored)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (b + k < e && sad4(w[k], tw) <= T) {
          if (park_ti0 < 0) { park_ti0 = ti; park_p0 = b + k; }
          else if (park_ti1 < 0) { park_ti1 = ti; park_p1 = b + k; }
          else decide(t, ti, b + k);
        }
      }
    }
  };
  const float* mine = reinterpret_cast<const float*>(s_strip[wave]) + 10 * lane;
  for (int u = 0; u < runs_per_wave && chunk0 + u < n_chunks; u += 2) {    // (wave-uniform)
    const int n2 = chunk0 + u + 2, n3 = chunk0 + u + 3;
    const RowRun ahead = run_load(tree, nt, n2 < n_chunks ? n2 : 0);       // (unconditional: see run_load)
    const RowRun ahead2 = run_load(tree, nt, n3 < n_chunks ? n3 : 0);
    const bool two = u + 1 < runs_per_wave && chunk0 + u + 1 < n_chunks;
    const Row10 tA = run_row(cur, s_strip[wave]);
    int bA, eA, bB, eB;
    lookup(mine, bA, eA);
    const Row10 tB = run_row(cur2, s_strip[wave]);
    lookup(mine, bB, eB);
    cur = ahead; cur2 = ahead2;
    const int tiA = 64 * (chunk0 + u) + lane, tiB = tiA + 64;
    visit(tA, tiA, bA, tiA < nt ? eA : 0);
    visit(tB, tiB, bB, (two && tiB < nt) ? eB : 0);
  }
  {
    const Row10 t0 = load_row(tree + 10 * (size_t)(park_ti0 >= 0 ? park_ti0 : 0));     // (unconditional: both rows in flight together)
    const Row10 t1 = load_row(tree + 10 * (size_t)(park_ti1 >= 0 ? park_ti1 : 0));
    if (park_ti0 >= 0) decide(t0, park_ti0, park_p0);
    if (park_ti1 >= 0) decide(t1, park_ti1, park_p1);
  }
}

// 2^log2p parts of 2^log2s words for a tree of nt points: at most ~0.39 * 2^log2s points per part on average (load factor)
struct HashPlan { int log2s, log2p, ib, qcap; };
#ifndef VO_HJ_MAXL
#define VO_HJ_MAXL 2
#endif
static bool hash_plan(int nt, HashPlan& p) {
  if (nt <= 0) return false;
  static const int fill[4] = {1600, 3200, 6400, 12800};               // log2s 12 .. 15
  // parts of at most 2^14 words: 64 KiB of table + 12 KiB of queues, so that TWO workgroups share a CU (with 2^15 words --
  // one workgroup per CU, half as many parts -- the step took 108 instead of ~60 us per 200 x 50k frames: its time is the
  // compare-and-swap chains' latency, which a second resident workgroup hides)
  int l = 0;
  while (l < VO_HJ_MAXL && nt > fill[l]) ++l;
  p.log2s = 12 + l;
  p.log2p = 0;
  while (p.log2p < 5 && (long long)nt > ((long long)fill[l] << p.log2p)) ++p.log2p;
  if ((long long)nt > ((long long)fill[l] << p.log2p)) return false;   // more than 204 800 points: the general search alone
  p.ib = 1;
  while ((1 << p.ib) < nt) ++p.ib;
  p.qcap = fill[l] + fill[l] / 2;                    // a part's queue: 1.5 x its average share (table load <= 0.59)
  return true;
}
static size_t hash_table_bytes(const HashPlan& p) { return align256(sizeof(unsigned) * ((size_t)1 << (p.log2s + p.log2p))); }
static size_t hash_queue_bytes(const HashPlan& p) { return align256(sizeof(uint2) * ((size_t)p.qcap << p.log2p)); }
static size_t hash_head_bytes(int n_frames) { return align256(sizeof(int) * (size_t)n_frames * (3 + HJ_MAXP)); }   // open-query counters, the sorted search's share of them (todo), sample masks, queue cursors
// workspace of the pass: the counters (zeroed per call), then the tables, then the queues
static size_t hash_list_bytes(int n_frames) { return align256(sizeof(int) * (size_t)n_frames * OPEN_MAX); }   // the open queries' indices
static size_t hash_ws_bytes(int nt, int n_frames) {
  HashPlan p;
  if (!hash_plan(nt, p)) return 0;
  return hash_head_bytes(n_frames) + hash_list_bytes(n_frames) + (hash_table_bytes(p) + hash_queue_bytes(p)) * (size_t)n_frames;
}
bool match_hash_supported(int nt, int n_frames) { (void)n_frames; HashPlan p; return hash_plan(nt, p); }
size_t match_hash_workspace_bytes(int nt, int n_frames) { return hash_ws_bytes(nt, n_frames); }

// (tree, qry, nt, nq): the two sets in their roles, or (set 1, set 2, capacities) with the per-frame sizes d_n1 / d_n2
static hipError_t launch_hash_first(hipStream_t st, const float* tree, int nt, const float* qry, int nq, float r2,
                                    unsigned long long* d_best, void* ws, int n_frames, size_t tree_stride, size_t qry_stride,
                                    size_t best_stride, const int* d_n1, const int* d_n2, int** d_unres_out, int32_t* d_out_pairs,
                                    size_t out_stride, int tree_is_1, bool automatic) {
  HashPlan p;
  const int nt_plan = d_n1 ? (nt > nq ? nt : nq) : nt;
  if (!hash_plan(nt_plan, p)) return hipErrorInvalidValue;
  HashArgs a;
  a.tree = tree; a.qry = qry; a.nt = nt; a.nq = nq;
  a.tree_stride = tree_stride; a.qry_stride = qry_stride; a.best_stride = best_stride;
  a.d_n1 = d_n1; a.d_n2 = d_n2;
  a.unres = static_cast<int*>(ws);
  a.sample_mask = reinterpret_cast<unsigned*>(a.unres + 2 * (size_t)n_frames);   // (unres + n_frames: todo[], written by open_collect_kernel)
  a.cursors = a.unres + 3 * (size_t)n_frames;
  a.open_list = reinterpret_cast<int*>(static_cast<char*>(ws) + hash_head_bytes(n_frames));
  a.tables = reinterpret_cast<unsigned*>(static_cast<char*>(ws) + hash_head_bytes(n_frames) + hash_list_bytes(n_frames));
  a.tables_stride = hash_table_bytes(p) / sizeof(unsigned);
  a.queues = reinterpret_cast<uint2*>(a.tables + a.tables_stride * (size_t)n_frames);
  a.qcap = p.qcap;
  a.sample_min = automatic ? HJ_SAMPLE_MIN : 1;
  a.hblocks = (nt_plan + 255) / 256;
  hipError_t e0 = hipMemsetAsync(ws, 0, sizeof(int) * (size_t)n_frames * (3 + HJ_MAXP), st);
  if (e0 != hipSuccess) return e0;
  a.best = d_best; a.r2 = r2;
  a.out_pairs = d_out_pairs; a.out_stride = out_stride; a.tree_is_1 = tree_is_1;
  a.n_frames = n_frames; a.log2p = p.log2p; a.ib = p.ib;
  const int q_cap = d_n1 ? (nt < nq ? nt : nq) : nq;       // ragged: either set may be the queries, never more than the smaller capacity
  a.qblocks = (q_cap + 256 * HJ_Q - 1) / (256 * HJ_Q);
  a.queue_stride = (size_t)p.qcap << p.log2p;
  a.seg_counts = a.qblocks * 256 * HJ_Q;                  // (4 (seg_counts + qblocks) <= 8 queue_stride: a queue holds 1.5 x the tree)
  if (4 * ((size_t)a.seg_counts + (size_t)a.qblocks) > 8 * a.queue_stride) { *d_unres_out = nullptr; return hipErrorNotSupported; }   // (unreachable while nq <= nt; the caller then runs the plain search)
  const dim3 gb(xcd_grid(1 << p.log2p, n_frames)), gp(xcd_grid(a.qblocks, n_frames)), tb(HJ_THREADS);
  hipLaunchKernelGGL(hash_rows_kernel, dim3(xcd_grid(a.hblocks, n_frames)), dim3(256), 0, st, a);
  switch (p.log2s) {
    case 12: hipLaunchKernelGGL(hash_table_kernel<12>, gb, tb, 0, st, a);
             hipLaunchKernelGGL(hash_probe_kernel<12>, gp, dim3(256), 0, st, a); break;
    case 13: hipLaunchKernelGGL(hash_table_kernel<13>, gb, tb, 0, st, a);
             hipLaunchKernelGGL(hash_probe_kernel<13>, gp, dim3(256), 0, st, a); break;
    case 14: hipLaunchKernelGGL(hash_table_kernel<14>, gb, tb, 0, st, a);
             hipLaunchKernelGGL(hash_probe_kernel<14>, gp, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL(hash_table_kernel<15>, gb, tb, 0, st, a);
             hipLaunchKernelGGL(hash_probe_kernel<15>, gp, dim3(256), 0, st, a); break;
  }
  hipLaunchKernelGGL(hash_open_kernel, dim3((unsigned)n_frames), dim3(256), 0, st, a);
  *d_unres_out = a.unres;
  return hipGetLastError();
}

// "did any frame of the call that just ran take the exact-duplicate pass?" (sample masks at the head of its workspace) into
// *d_out: what the host reads -- without waiting, capi.hip -- to decide whether the next calls run the pass at all
__global__ void hash_hint_kernel(const unsigned* masks, int n_frames, int sample_min, int* out) {
  int any = 0;
  for (int f = threadIdx.x; f < n_frames; f += blockDim.x) any |= __popc(masks[f]) >= sample_min ? 1 : 0;
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) *out = any;
}
// the same question asked of a call that ran WITHOUT the pass: eight evenly spread queries per frame -- does any of them sit at
// distance exactly 0 from its match (a bitwise copy, for rows the pass would take)?  Reads the keys the search left.
__global__ void best_hint_kernel(const unsigned long long* best, size_t best_stride, int nq_cap, const int* d_n1, const int* d_n2,
                                 int n_frames, int* out) {
  int any = 0;
  for (int i = threadIdx.x; i < n_frames * 8; i += blockDim.x) {
    const int f = i >> 3, k = i & 7;
    int nq = nq_cap;
    if (d_n1 && d_n2) { const int a = d_n1[f], b = d_n2[f]; nq = a < b ? a : b; nq = nq < 0 ? 0 : (nq > nq_cap ? nq_cap : nq); }
    if (nq <= 0) continue;
    const unsigned long long key = best[(size_t)f * best_stride + (size_t)(k * (nq >> 3))];
    any |= ((unsigned)(key >> 32) == 0u && (unsigned)key != 0xffffffffu) ? 1 : 0;
  }
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) *out = any;
}
hipError_t launch_match_hint_from_best(hipStream_t st, const unsigned long long* d_best, size_t best_stride, int nq_cap,
                                       const int* d_n1, const int* d_n2, int n_frames, int* d_out) {
  hipLaunchKernelGGL(best_hint_kernel, dim3(1), dim3(256), 0, st, d_best, best_stride, nq_cap, d_n1, d_n2, n_frames, d_out);
  return hipGetLastError();
}
hipError_t launch_match_hint(hipStream_t st, const void* d_prune_ws, int n_frames, int* d_out) {
  const unsigned* masks = reinterpret_cast<const unsigned*>(static_cast<const int*>(d_prune_ws) + 2 * (size_t)n_frames);
  hipLaunchKernelGGL(hash_hint_kernel, dim3(1), dim3(256), 0, st, masks, n_frames, HJ_SAMPLE_MIN, d_out);
  return hipGetLastError();
}

// compute units of the device the running call targets (launch_match_batch sets it from its n_cu argument; the radius search,
// which has none, keeps the last value or MI355X's 256)
static thread_local int t_n_cu = 256;
static hipError_t launch_cells_sort(hipStream_t st, CellArgs& a, const float* tree, int nt, const float* qry, int nq,
                                    float radius, float r2, unsigned long long* d_best, void* ws, int n_frames,
                                    size_t tree_stride, size_t qry_stride, size_t best_stride, const int* d_n1 = nullptr,
                                    const int* d_n2 = nullptr, const int* d_unres = nullptr, int* d_todo = nullptr, const int* d_open_list = nullptr) {
  a.tree = tree; a.qry = qry; a.nt = nt; a.nq = nq;
  a.d_n1 = d_n1; a.d_n2 = d_n2; a.unres = d_unres; a.todo = nullptr; a.open_div = 0; a.open_list = nullptr;
  a.ws = static_cast<char*>(ws);
  if (d_n1) { const int cap = nt > nq ? nt : nq; a.w = cell_ws_layout(cap, cap); }   // ragged: either set may play either role
  else a.w = cell_ws_layout(nt, nq);
  a.ws_stride = a.w.total; a.tree_stride = tree_stride; a.qry_stride = qry_stride; a.best_stride = best_stride;
  a.n_frames = n_frames;
  a.tb = cell_slices(d_n1 ? (nt > nq ? nt : nq) : nt); a.qb = cell_slices(d_n1 ? (nt > nq ? nt : nq) : nq);
  a.radius = radius; a.r2 = r2; a.best = d_best;
  a.rs_offsets = nullptr; a.rs_indices = nullptr; a.rs_capacity = 0;
  const unsigned Z = (unsigned)n_frames;
  hipLaunchKernelGGL(cell_bounds_kernel, dim3(Z), dim3(1024), 0, st, a);
  if (d_unres && d_todo && d_open_list) {
    // frames with few open queries: the tree streamed past them; the sorted search below then sees todo[] instead of unres[]
    static const int open_div = [] { const char* e = getenv("VO_MATCH_OPEN_DIV"); return e ? atoi(e) : OPEN_DIV_DEFAULT; }();
    a.todo = d_todo; a.open_div = open_div; a.open_list = d_open_list;
    const int big = d_n1 ? (nt > nq ? nt : nq) : nt;
    // ~4 workgroups per CU over the call, so that the staging round trip at a workgroup's start is a small part of it
    const int runs = (big + 63) / 64;
    int tblocks = (4 * t_n_cu + n_frames - 1) / n_frames;
    const int tb_max = (runs + OPEN_SCAN_WAVES - 1) / OPEN_SCAN_WAVES;
    tblocks = tblocks > tb_max ? tb_max : tblocks;
    const int runs_per_wave = (runs + tblocks * OPEN_SCAN_WAVES - 1) / (tblocks * OPEN_SCAN_WAVES);
    hipLaunchKernelGGL(open_collect_kernel, dim3(Z), dim3(OPEN_THREADS), 0, st, a);
    hipLaunchKernelGGL(open_scan_kernel, dim3(xcd_grid(tblocks, n_frames)), dim3(OPEN_SCAN_THREADS), 0, st, a, tblocks, runs_per_wave);
    a.unres = d_todo;
  }
  hipLaunchKernelGGL(cell_place_kernel, dim3(xcd_grid(a.tb + a.qb, n_frames)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(cell_offsets_kernel, dim3(2, 1, Z), dim3(HCPAD), 0, st, a);
  hipLaunchKernelGGL(cell_fine_kernel, dim3(xcd_grid(HCOARSE / FG, n_frames)), dim3(256), 0, st, a);
  return hipGetLastError();
}

static hipError_t launch_match_cells(hipStream_t st, const float* tree, int nt, const float* qry, int nq,
                                     float radius, float r2, unsigned long long* d_best, void* ws, int n_frames,
                                     size_t tree_stride, size_t qry_stride, size_t best_stride, const int* d_n1 = nullptr,
                                     const int* d_n2 = nullptr, const int* d_unres = nullptr, int* d_todo = nullptr, const int* d_open_list = nullptr) {
  CellArgs a;
  hipError_t e = launch_cells_sort(st, a, tree, nt, qry, nq, radius, r2, d_best, ws, n_frames, tree_stride, qry_stride,
                                   best_stride, d_n1, d_n2, d_unres, d_todo, d_open_list);
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
                              int variant_in, const int* d_n1, const int* d_n2) {
  // variants 4 / 5: the exact-duplicate pass first, then variant 2 / 3 for the queries it left open (its workspace comes
  // first in d_prune_ws)
  if (n_cu > 0) t_n_cu = n_cu;
  const bool automatic = (variant_in & MATCH_VARIANT_AUTO) != 0;       // picked by the automatic rule, not forced (vo_match_set_mode)
  variant_in &= ~MATCH_VARIANT_AUTO;
  const bool hash_first = variant_in >= 4 && d_prune_ws != nullptr;
  const int variant = variant_in >= 4 ? variant_in - 2 : variant_in;
  int* d_unres = nullptr;
  if (d_n1 && d_n2) {
    // ragged frames: every frame its own sizes and roles; full scan or (variant 3) the cell-hash search.
    // q = min(n1, n2) is the room per frame in d_best / d_out_pairs.
    const int q = n1 < n2 ? n1 : n2;
    const float r2 = radius * radius;
    if (q > 0 && variant == 3 && d_prune_ws) {
      void* ws = d_prune_ws;
      if (hash_first) {
        hipError_t eh = launch_hash_first(st, d_a1, n1, d_a2, n2, r2, d_best, d_prune_ws, n_frames, a1_stride, a2_stride, (size_t)q,
                                          d_n1, d_n2, &d_unres, q > SMALL_COMPACT ? d_out_pairs : nullptr, out_stride, 1, automatic);
        if (eh == hipErrorNotSupported) d_unres = nullptr;        // sizes the pass does not take: the plain search answers every query
        else if (eh != hipSuccess) return eh;
        ws = static_cast<char*>(d_prune_ws) + hash_ws_bytes(n1 > n2 ? n1 : n2, n_frames);
      }
      // the cell-hash search with per-frame sizes and roles (workspace laid out for max(n1, n2) in both roles)
      hipError_t ec = launch_match_cells(st, d_a1, n1, d_a2, n2, radius, r2, d_best, ws, n_frames, a1_stride, a2_stride,
                                         (size_t)q, d_n1, d_n2, d_unres, d_unres ? d_unres + n_frames : nullptr,
                                         d_unres ? reinterpret_cast<const int*>(static_cast<char*>(d_prune_ws) + hash_head_bytes(n_frames)) : nullptr);
      if (ec != hipSuccess) return ec;
    } else if (q > 0) {
      hipLaunchKernelGGL(match_init_kernel, dim3((q + 255) / 256, 1, (unsigned)n_frames), dim3(256), 0, st, d_best, q, r2, (size_t)q);
      const int big = n1 > n2 ? n1 : n2;
      const int qblocks = (q + MB * QPT - 1) / (MB * QPT);      // either set may be the queries, but never more than q of them
      int want = (8 * (n_cu > 0 ? n_cu : 256) + qblocks * n_frames - 1) / (qblocks * n_frames);
      const int tiles = (big + TILE - 1) / TILE;
      if (want > tiles) want = tiles;
      if (want < 1) want = 1;
      const int chunk = ((tiles + want - 1) / want) * TILE;
      const int nchunks = (big + chunk - 1) / chunk;
      hipLaunchKernelGGL(match_kernel<true>, dim3(qblocks, nchunks, (unsigned)n_frames), dim3(MB), 0, st, d_a1, n1, d_a2, n2, chunk, r2,
                         d_best, a1_stride, a2_stride, (size_t)q, d_n1, d_n2);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_match_compact(st, d_best, q, 1, d_out_pairs, d_n_out, d_scratch, n_frames, (size_t)q, out_stride, d_n1, d_n2, n1, n2, d_unres);
  }
  const int tree_is_1 = n1 >= n2;                 // vo_complete.cpp:15-20 (ties: a1 is the tree)
  const float* tree = tree_is_1 ? d_a1 : d_a2;
  const float* qry = tree_is_1 ? d_a2 : d_a1;
  const size_t ts = tree_is_1 ? a1_stride : a2_stride, qs = tree_is_1 ? a2_stride : a1_stride;
  const int nt = tree_is_1 ? n1 : n2, nq = tree_is_1 ? n2 : n1;
  const float r2 = radius * radius;
  const size_t best_stride = n_frames > 1 ? (size_t)nq : 0;
  const unsigned Z = (unsigned)n_frames;
  void* ws = d_prune_ws;
  if (hash_first && nq > 0 && nt > 0 && (variant == 2 || variant == 3)) {
    hipError_t eh = launch_hash_first(st, tree, nt, qry, nq, r2, d_best, d_prune_ws, n_frames, n_frames > 1 ? ts : 0,
                                      n_frames > 1 ? qs : 0, best_stride, nullptr, nullptr, &d_unres,
                                      nq > SMALL_COMPACT ? d_out_pairs : nullptr, n_frames > 1 ? out_stride : 0, tree_is_1, automatic);
    if (eh == hipErrorNotSupported) d_unres = nullptr;            // sizes the pass does not take: the plain search answers every query
    else if (eh != hipSuccess) return eh;
    ws = static_cast<char*>(d_prune_ws) + hash_ws_bytes(nt, n_frames);
  }
  if (nq > 0 && nt > 0 && d_prune_ws && variant == 3) {
    hipError_t ep = launch_match_cells(st, tree, nt, qry, nq, radius, r2, d_best, ws, n_frames, ts, qs,
                                       best_stride, nullptr, nullptr, d_unres, d_unres ? d_unres + n_frames : nullptr,
                                       d_unres ? reinterpret_cast<const int*>(static_cast<char*>(d_prune_ws) + hash_head_bytes(n_frames)) : nullptr);
    if (ep != hipSuccess) return ep;
  } else if (nq > 0 && nt > 0 && d_prune_ws) {
    hipError_t ep = launch_match_pruned(st, tree, nt, qry, nq, radius, r2, d_best, ws, n_cu, n_frames, ts, qs,
                                        best_stride, d_unres);
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
      hipLaunchKernelGGL(match_kernel<false>, dim3(qblocks, nchunks, Z), dim3(MB), 0, st, tree, nt, qry, nq, chunk, r2, d_best,
                         n_frames > 1 ? ts : 0, n_frames > 1 ? qs : 0, best_stride, nullptr, nullptr);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_match_compact(st, d_best, nq, tree_is_1, d_out_pairs, d_n_out, d_scratch, n_frames, best_stride,
                              out_stride, nullptr, nullptr, 0, 0, d_unres);
}

hipError_t launch_match(hipStream_t st, const float* d_a1, int n1, const float* d_a2, int n2,
                        float radius, int32_t* d_out_pairs, int* d_n_out,
                        unsigned long long* d_best, int* d_scratch, int n_cu, void* d_prune_ws, int variant) {
  return launch_match_batch(st, d_a1, n1, 0, d_a2, n2, 0, radius, d_out_pairs, 0, d_n_out, d_best, d_scratch, n_cu,
                            d_prune_ws, 1, variant, nullptr, nullptr);
}

}  // namespace vo
