// map.hip -- the map of vo_complete on the device: PointCloudVector<3>::update (PointCloud.h:52-66) and the `history`
// chain that moves every frame's triangulated cloud into the first camera's frame (vo_complete.cpp:145-147,175-176).
//
// The reference's update is a sequential double loop: for every point of the cloud, in order, the FIRST map entry whose
// appearance compares equal (operator== on ten floats: -0 == +0, a NaN equals nothing) gets the point; with no such entry
// the (point, appearance) pair is appended -- and is itself found by later points of the same cloud.  What that leaves:
//   * one entry per class of equal appearances, holding the appearance BITS of the class's first occurrence ever
//     and the point of its LAST occurrence so far;
//   * new classes appended in the order of their first occurrence in the cloud;
//   * rows with a NaN always appended, each one, never found.
// Here the classes live in an open-addressing table in device memory keyed by a hash of the canonical row (-0 -> +0), one
// 64-bit word per slot: (tag << 32) | entry.  One update = four launches over the cloud, no host round trip:
//   map_probe_kernel   every cloud row finds or claims its class's slot.  A row that claims writes (tag, M + i) -- M the
//                      map's size, i its index in the cloud -- and equal rows lower that to the smallest i by atomicMin
//                      (an entry < M, i.e. a class the map already holds, is smaller than any of them); the class's last
//                      occurrence is kept beside it by atomicMax.  Slot per row -> scratch.
//   map_flag_kernel    "am I the first occurrence of a new class?" per row; per-workgroup counts and in-group ranks.
//   map_scan_kernel    one workgroup: offsets of the workgroups, new size.
//   map_commit_kernel  first occurrences write their appearance at M + rank; last occurrences write the (moved) point
//                      into the class's entry and turn the slot's provisional M + i into the entry's index.
// The result does not depend on the order in which rows reach the table: min / max over cloud indices decide.
#include "vo_internal.h"

namespace vo {

constexpr int MB = 256;                                   // rows per workgroup
constexpr unsigned long long MAP_EMPTY = ~0ull;

struct MapArgs {
  float* pts;                  // [cap][3]
  float* app;                  // [cap][10]
  unsigned long long* table;   // [tcap] (tag << 32) | entry, MAP_EMPTY when free
  int* last;                   // [tcap] last cloud index of the slot's class in the running update, -1 otherwise
  unsigned tmask;              // tcap - 1 (a power of two)
  int cap;                     // entries the arrays hold
  int* hdr;                    // [0] size  [1] base of the running update  [2] rows of the running update  [3] rows dropped for lack of room
  const float* c_xyz;          // the cloud
  const float* c_app;
  int n_max;
  const int* d_n;              // or null
  const float* T16;            // isometry applied to every cloud point (device, column-major 4x4) or null
  int* slot;                   // [n_max] scratch: slot of row i, -1 for a NaN row
  int* rank;                   // [n_max] scratch: rank of a first occurrence inside its workgroup
  int* counts;                 // [nb + 1] scratch: per-workgroup counts -> offsets
  int nb;
};

struct Row { float v[10]; };

__device__ __forceinline__ Row map_load_row(const float* app, size_t i) {
  const float2* p = reinterpret_cast<const float2*>(app) + 5 * i;
  Row r;
#pragma unroll
  for (int k = 0; k < 5; ++k) { const float2 t = p[k]; r.v[2 * k] = t.x; r.v[2 * k + 1] = t.y; }
  return r;
}
__device__ __forceinline__ bool map_row_has_nan(const Row& r) {
  bool nan = false;
#pragma unroll
  for (int k = 0; k < 10; ++k) nan |= r.v[k] != r.v[k];
  return nan;
}
// operator== on ten floats (PointCloud.h:56) for rows without NaN
__device__ __forceinline__ bool map_rows_equal(const Row& a, const Row& b) {
  bool eq = true;
#pragma unroll
  for (int k = 0; k < 10; ++k) eq &= a.v[k] == b.v[k];
  return eq;
}
__device__ __forceinline__ unsigned map_rotl(unsigned x, int r) { return (x << r) | (x >> (32 - r)); }
// hash of the canonical row: -0 hashes as +0, so that rows equal under == share a hash
__device__ __forceinline__ unsigned map_hash(const Row& r) {
  unsigned x = 0x9e3779b9u;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const unsigned w = r.v[k] == 0.f ? 0u : __float_as_uint(r.v[k]);
    x = (k & 1) ? map_rotl(x, 7) + w : map_rotl(x, 11) ^ w;
  }
  x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
  return x;
}
__device__ __forceinline__ int map_rows(const MapArgs& a) {
  int n = a.n_max;
  if (a.d_n) { const int m = *a.d_n; n = m < n ? (m < 0 ? 0 : m) : n; }
  return n;
}

__global__ __launch_bounds__(MB) void map_probe_kernel(MapArgs a) {
  const int n = map_rows(a);
  const int M = a.hdr[0];
  const int i = blockIdx.x * MB + threadIdx.x;
  if (i >= n) return;
  const Row r = map_load_row(a.c_app, (size_t)i);
  if (map_row_has_nan(r)) { a.slot[i] = -1; return; }
  const unsigned h = map_hash(r);
  const unsigned long long mine = ((unsigned long long)h << 32) | (unsigned)(M + i);
  unsigned s = (h * 0x9e3779b1u) & a.tmask;                 // home slot: other bits than the tag's comparison relies on
  int found = -2;
  for (unsigned probes = 0; probes <= a.tmask; ++probes, s = (s + 1) & a.tmask) {
    unsigned long long w = a.table[s];
    if (w == MAP_EMPTY) {
      w = atomicCAS(&a.table[s], MAP_EMPTY, mine);
      if (w == MAP_EMPTY) { found = (int)s; break; }        // claimed
    }
    if ((unsigned)(w >> 32) != h) continue;                 // another class lives here
    const unsigned e = (unsigned)w;
    const Row o = e < (unsigned)M ? map_load_row(a.app, (size_t)e) : map_load_row(a.c_app, (size_t)(e - (unsigned)M));
    if (!map_rows_equal(r, o)) continue;                    // same tag, another row
    // my class: from here on the slot only ever holds members of it, so the minimum keeps the map's entry if there is
    // one and the first occurrence in the cloud otherwise
    atomicMin(&a.table[s], mine);
    found = (int)s;
    break;
  }
  a.slot[i] = found;                                        // (-2: the table is full -- the host keeps it at most half full)
  if (found >= 0) atomicMax(&a.last[found], i);
}

__global__ __launch_bounds__(MB) void map_flag_kernel(MapArgs a) {
  __shared__ int s_wave[MB / 64];
  const int n = map_rows(a);
  const int M = a.hdr[0];
  const int i = blockIdx.x * MB + threadIdx.x;
  bool first = false;
  if (i < n) {
    const int s = a.slot[i];
    first = s == -1 || (s >= 0 && (unsigned)a.table[s] == (unsigned)(M + i));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(first);
  if (lane == 0) s_wave[wave] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < MB / 64; ++w) { const int c = s_wave[w]; if (w < wave) off += c; tot += c; }
  if (i < a.n_max) a.rank[i] = first ? off + __popcll(m & ((1ull << lane) - 1ull)) : -1;
  if (threadIdx.x == 0) a.counts[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void map_scan_kernel(MapArgs a) {
  __shared__ int s_w[16];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < a.nb; base += 1024) {
    const int j = base + tid;
    const int v = j < a.nb ? a.counts[j] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int c = s_w[w]; if (w < wave) woff += c; tot += c; }
    const int carry = s_carry;
    if (j < a.nb) a.counts[j] = carry + woff + incl - v;
    __syncthreads();
    if (tid == 0) s_carry = carry + tot;
    __syncthreads();
  }
  if (tid == 0) {
    const int M = a.hdr[0];
    int add = s_carry;
    int dropped = 0;
    if (M + add > a.cap) { dropped = M + add - a.cap; add = a.cap - M; }      // (the host grows the arrays before this can happen)
    a.hdr[1] = M;
    a.hdr[2] = map_rows(a);
    a.hdr[3] += dropped;
    a.hdr[0] = M + add;
  }
}

__global__ __launch_bounds__(MB) void map_commit_kernel(MapArgs a) {
  const int n = a.hdr[2];
  const int M = a.hdr[1];
  const int i = blockIdx.x * MB + threadIdx.x;
  if (i >= n) return;
  const int s = a.slot[i];
  if (s == -2) return;
  const int rk = a.rank[i];
  if (rk >= 0) {                                            // first occurrence of a new class (or a NaN row): its appearance, bit for bit
    const int e = M + a.counts[blockIdx.x] + rk;
    if (e < a.cap) {
      const float2* src = reinterpret_cast<const float2*>(a.c_app) + 5 * (size_t)i;
      float2* dst = reinterpret_cast<float2*>(a.app) + 5 * (size_t)e;
#pragma unroll
      for (int k = 0; k < 5; ++k) dst[k] = src[k];
    }
  }
  int e;                                                    // the entry this row's point goes to, if it is its class's last occurrence
  if (s == -1) {
    e = M + a.counts[blockIdx.x] + rk;
  } else {
    if (a.last[s] != i) return;
    const unsigned idx = (unsigned)a.table[s];
    if (idx < (unsigned)M) {
      e = (int)idx;
    } else {
      const int f = (int)(idx - (unsigned)M);               // the class's first occurrence in this cloud
      e = M + a.counts[f / MB] + a.rank[f];
      // the slot now names the entry (only this thread touches the slot in this launch)
      a.table[s] = (a.table[s] & 0xffffffff00000000ull) | (unsigned)e;
    }
    a.last[s] = -1;
  }
  if (e >= a.cap) return;
  float x = a.c_xyz[3 * (size_t)i], y = a.c_xyz[3 * (size_t)i + 1], z = a.c_xyz[3 * (size_t)i + 2];
  if (a.T16) {                                              // history * triangulated_pc (vo_complete.cpp:175, PointCloud.h:80)
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = a.T16[k];
    const Pose X = pose_from_T16(t);
    const float px = x, py = y, pz = z;
    pose_apply(X, px, py, pz, x, y, z);
  }
  a.pts[3 * (size_t)e] = x; a.pts[3 * (size_t)e + 1] = y; a.pts[3 * (size_t)e + 2] = z;
}

// rebuilds the table from the entries (after the arrays grew): entry j claims or joins its class's slot; the smallest j stays
__global__ __launch_bounds__(MB) void map_rehash_kernel(MapArgs a) {
  const int M = a.hdr[0];
  const int j = blockIdx.x * MB + threadIdx.x;
  if (j >= M) return;
  const Row r = map_load_row(a.app, (size_t)j);
  if (map_row_has_nan(r)) return;
  const unsigned h = map_hash(r);
  const unsigned long long mine = ((unsigned long long)h << 32) | (unsigned)j;
  unsigned s = (h * 0x9e3779b1u) & a.tmask;
  for (unsigned probes = 0; probes <= a.tmask; ++probes, s = (s + 1) & a.tmask) {
    unsigned long long w = a.table[s];
    if (w == MAP_EMPTY) {
      w = atomicCAS(&a.table[s], MAP_EMPTY, mine);
      if (w == MAP_EMPTY) return;
    }
    if ((unsigned)(w >> 32) != h) continue;
    if (!map_rows_equal(r, map_load_row(a.app, (size_t)(unsigned)w))) continue;
    atomicMin(&a.table[s], mine);
    return;
  }
}

// history <- X^-1 (reset != 0, vo_complete.cpp:146) or history * X^-1 (vo_complete.cpp:176): Isometry3f arithmetic in float,
// Eigen's order (vo_math.h: pose_inverse, pose_mul), one thread
__global__ void map_history_kernel(float* hist16, const float* X16, int reset) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float t[16];
  for (int k = 0; k < 16; ++k) t[k] = X16[k];
  const Pose Xi = pose_inverse(pose_from_T16(t));
  Pose H = Xi;
  if (!reset) {
    for (int k = 0; k < 16; ++k) t[k] = hist16[k];
    H = pose_mul(pose_from_T16(t), Xi);
  }
  pose_to_T16(H, t);
  for (int k = 0; k < 16; ++k) hist16[k] = t[k];
}

// map <- T * map (vo_complete.cpp:183), in place
__global__ __launch_bounds__(256) void map_transform_kernel(MapArgs a, Pose T) {
  const int M = a.hdr[0];
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) {
    float* p = a.pts + 3 * (size_t)j;
    float x, y, z;
    pose_apply(T, p[0], p[1], p[2], x, y, z);
    p[0] = x; p[1] = y; p[2] = z;
  }
}

size_t map_scratch_ints(int n_max) { return 2 * (size_t)n_max + (size_t)((n_max + MB - 1) / MB) + 8; }

static MapArgs map_args(const MapDev& m, const float* c_xyz, const float* c_app, int n_max, const int* d_n, const float* T16,
                        int* scratch) {
  MapArgs a;
  a.pts = m.pts; a.app = m.app; a.table = m.table; a.last = m.last; a.tmask = m.tcap - 1u; a.cap = m.cap; a.hdr = m.hdr;
  a.c_xyz = c_xyz; a.c_app = c_app; a.n_max = n_max; a.d_n = d_n; a.T16 = T16;
  a.nb = (n_max + MB - 1) / MB;
  a.slot = scratch; a.rank = scratch + n_max; a.counts = scratch + 2 * (size_t)n_max;
  return a;
}

hipError_t launch_map_update(hipStream_t st, const MapDev& m, const float* c_xyz, const float* c_app, int n_max, const int* d_n,
                             const float* d_T16, int* d_scratch) {
  if (n_max <= 0) return hipSuccess;
  const MapArgs a = map_args(m, c_xyz, c_app, n_max, d_n, d_T16, d_scratch);
  hipLaunchKernelGGL(map_probe_kernel, dim3(a.nb), dim3(MB), 0, st, a);
  hipLaunchKernelGGL(map_flag_kernel, dim3(a.nb), dim3(MB), 0, st, a);
  hipLaunchKernelGGL(map_scan_kernel, dim3(1), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(map_commit_kernel, dim3(a.nb), dim3(MB), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_map_rehash(hipStream_t st, const MapDev& m, int size_bound) {
  hipError_t e = hipMemsetAsync(m.table, 0xff, sizeof(unsigned long long) * (size_t)m.tcap, st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(m.last, 0xff, sizeof(int) * (size_t)m.tcap, st);
  if (e != hipSuccess || size_bound <= 0) return e;
  const MapArgs a = map_args(m, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(map_rehash_kernel, dim3((size_bound + MB - 1) / MB), dim3(MB), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_map_history(hipStream_t st, float* d_hist16, const float* d_X16, int reset) {
  hipLaunchKernelGGL(map_history_kernel, dim3(1), dim3(64), 0, st, d_hist16, d_X16, reset);
  return hipGetLastError();
}

hipError_t launch_map_transform(hipStream_t st, const MapDev& m, const Pose& T, int size_bound) {
  if (size_bound <= 0) return hipSuccess;
  const MapArgs a = map_args(m, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
  int grid = (size_bound + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(map_transform_kernel, dim3(grid), dim3(256), 0, st, a, T);
  return hipGetLastError();
}

}  // namespace vo
