// vo_internal.h -- declarations shared by the translation units of libvo_hip.so
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "vo_math.h"

// records the message vo_last_error() returns on this thread and hands `code` back (capi.hip)
int vo_fail(int code, const char* fmt, ...);
extern "C" __attribute__((visibility("hidden"))) int vo_ctx_capturing(struct vo_ctx* ctx);   // 1 while a graph capture is in progress on the context (capi.hip)
extern "C" __attribute__((visibility("hidden"))) int vo_ctx_alive(struct vo_ctx* ctx);       // 0 once the context has been destroyed (handles may outlive it)
extern "C" __attribute__((visibility("hidden"))) unsigned long long vo_ctx_id(struct vo_ctx* ctx);   // unique id of a live context (0: not alive): addresses get recycled, ids do not

namespace vo {

// ---- PICP ------------------------------------------------------------------
#ifndef VO_PICP_BLOCK
#define VO_PICP_BLOCK 256
#endif
constexpr int PICP_BLOCK = VO_PICP_BLOCK;   // threads per workgroup of the single-problem kernels (256, 512 or 1024: DESIGN.md section 4.1)
constexpr int PICP_PSTRIDE = 32;      // floats per workgroup partial (NACC padded)
constexpr int PICP_MAX_BLOCKS = 2048; // grid cap (grid-stride beyond it)
#ifndef VO_PICP_REPLICAS
#define VO_PICP_REPLICAS 2
#endif
// Copies of the workgroup partial rows.  Every workgroup of a launch reads ALL the rows the previous launch wrote -- 196
// workgroups fetching the same 25 KB at the same instant from the memory side (the rows come from all eight XCDs): with two
// copies (a writer stores its row twice, a reader takes copy blockIdx.x % 2) the round of the 50k headline takes 4.48-4.50
// instead of 4.57-4.58 us; four copies 4.50, eight 4.55 (the stores), three 4.63 (the modulo).
constexpr int PICP_REPLICAS = VO_PICP_REPLICAS;
// Slots of the round-to-round hand-off (workgroup partial rows, pose): round `it` reads slot (it - 1) % PICP_SLOTS and writes
// slot it % PICP_SLOTS.  Two would do for rounds enqueued one at a time; more let vo_picp_one_round enqueue up to PICP_SLOTS - 2
// rounds AHEAD of its caller (capi.hip) without touching the slot of the last counted round.  Sixteen, with windows of eight
// rounds: consecutive windows of a loop then start at two alternating slots only, i.e. TWO launch graphs serve a loop.
#ifndef VO_PICP_SLOTS
#define VO_PICP_SLOTS 16
#endif
constexpr int PICP_SLOTS = VO_PICP_SLOTS;     // a power of two
constexpr int PICP_BATCH_BLOCK = 768;   // 12 waves: 3 per SIMD, 168 VGPRs each (room for load double-buffering)
constexpr int PICP_BATCH_LDS_TRIPS = 2;   // trips of the batched solver held in LDS across rounds (2 x 60 KiB)

// Solver parameters, resident in device memory so that a captured graph of
// iteration launches stays valid when the camera / threshold / count change.
struct PicpParams {
  CamK cam;
  float thr;
  float damping;
  int keep_outliers;
  int n_corr;
};

// Solver state in device memory.
struct PicpState {
  float pose[PICP_SLOTS][12];   // ring (slot = round % PICP_SLOTS; slot 0 also holds the finished pose): R (col-major 3x3) then t
  float H[36];         // last round, damping included (col-major)
  float b[6];
  float chi_in, chi_out;
  int n_in;
  int n_bad;           // correspondences whose indices were out of range (dropped)
  float T16[16];       // pose after the last solve as a column-major 4x4 (written by the finish launch)
#ifdef VO_STAMPS
  // diagnostic build only (make STAMPS=1 -> libvo_hip_stamps.so, tools/stamp_rounds.py):
  // s_memtime at phase boundaries of workgroup 0, per round
  unsigned long long stamps[128][8];
#endif
};

// packed correspondences: five SoA arrays of `cap` floats each (x,y,z,u,v)
struct PackedCorr {
  float* base;
  size_t cap;
  __host__ __device__ const float* arr(int k) const { return base + (size_t)k * cap; }
  __host__ __device__ float* arr(int k) { return base + (size_t)k * cap; }
};

// d_T0 (may be null): column-major 4x4 in device memory that becomes the solver's pose
hipError_t launch_picp_pack(hipStream_t st, const int32_t* d_pairs, const int* d_n, int n_max,
                            const float* d_world, int n_world, const float* d_meas, int n_meas,
                            PackedCorr pk, PicpParams* d_params, PicpState* d_state, const float* d_T0);

// Enqueue n_iters Gauss-Newton rounds (n_iters+1 launches).  d_partials holds
// PICP_REPLICAS * PICP_SLOTS * round_up(grid,256) * PICP_PSTRIDE floats, zero-initialised.
hipError_t launch_picp_rounds(hipStream_t st, const PicpParams* d_params, PicpState* d_state,
                              PackedCorr pk, float* d_partials, int grid, int n_iters, bool pinhole,
                              bool keep_outliers);

int picp_grid_for(int n_corr, int n_cu);

// A chain of rounds whose finishing launch is deferred (vo_picp_one_round): round `it` is one launch; launch_picp_finish
// closes a chain of n_rounds.  picp_rounds_chain(grid): false when launch_picp_rounds runs this size as ONE launch anyway
// (a single workgroup: picp_small_kernel), which leaves nothing to defer.
bool picp_rounds_chain(int grid);
hipError_t launch_picp_chain_round(hipStream_t st, const PicpParams* d_params, PicpState* d_state, PackedCorr pk,
                                   float* d_partials, int grid, int it, bool pinhole, bool keep_outliers);
hipError_t launch_picp_finish(hipStream_t st, const PicpParams* d_params, PicpState* d_state, PackedCorr pk,
                              float* d_partials, int grid, int n_rounds);

// Reference-order solver (bit-identical to the reference's scalar arithmetic): n_iters rounds in one launch of one
// workgroup; reads the packed correspondences and P->n_corr, leaves pose / T16 / H / b / statistics in *d_state.
hipError_t launch_picp_exact(hipStream_t st, const PicpParams* d_params, PicpState* d_state, PackedCorr pk,
                             int n_iters);

struct BatchArgs {
  CamK cam;
  float thr, damping;
  int keep_outliers;
  int n_iters;
  int n_problems;
  const float* world; size_t world_stride;   // strides in points
  const float* meas; size_t meas_stride;
  const int32_t* pairs; size_t pairs_stride; // stride in pairs
  const int* n_pairs;
  const float* T0;     // n_problems x 16 or null
  float* T_out;        // n_problems x 16
  float* stats_out;    // n_problems x 4 or null
  float* packed;       // workspace: n_problems x 5 x cap floats
  size_t cap;          // per-array capacity (multiple of 4)
  int n_world, n_meas; // bounds for index checks (per problem)
  // launch-per-round form only (a few problems, many workgroups each): per-problem state and partial buffers
  PicpState* states;   // n_problems, or null (one-workgroup-per-problem kernel)
  float* partials;     // n_problems x 2 x round_up(grid,256) x PICP_PSTRIDE floats, zero-padded rows
  const PicpParams* params;   // device copy of (cam, thr, damping, keep_outliers); n_corr unused
  int grid;            // workgroups per problem
  int pack_gx;         // workgroups per problem of the gather pass
  int* n_bad;          // n_problems counters (zeroed by the caller): correspondences dropped for a bad index, or null
  int exact;           // reference-order form (picp_exact_kernel): one workgroup per problem, sequential sums
  const float* X_world;  // n_problems x 16 (column-major) or null: the gather applies X * p to every world point it fetches
                         //   (X_curr * triangulated_pc of vo_complete.cpp:159 without the pass that writes the moved cloud)
  int prepacked;         // the packed arrays are already filled (the join's writing pass gathered through its own pairs:
                         //   launch_join_batch with a sink): no gather pass
  // fewer problems than CUs (picp_batch_shared_kernel): the workgroups beyond n_problems take trips off the problems' own
  unsigned long long* help_words;   // picp_help_words(n_problems, n_cu) tagged words, zeroed by launch_picp_batch; null: form off
  int help_grid;         // workgroups of the launch (n_problems + helpers), 0: form off
  int help_rows;         // partial rows in help_words
  int help_absent;       // test hook (VO_PICP_HELP_ABSENT=1): the helper waves leave at once, the homes stand in for all of them
  int help_keep, help_g, help_slack10;   // 0: the kernel's own choice (experiments: VO_PICP_HELP_KEEP / _G / _SLACK): trips a home keeps,
                         //   wave-trips per chunk, a helper's overhead per round in tenths of a trip
};
hipError_t launch_picp_batch(hipStream_t st, const BatchArgs& a);
// the shared form: whether it serves this call, and what it needs (rows of 32 words, then 16 words per problem)
bool picp_batch_shares(int n_problems, size_t cap, int n_iters, int n_cu);
int picp_help_rows(int n_problems, int n_cu);
void picp_help_args(BatchArgs& a, unsigned long long* words, int n_cu);     // words null: form off
inline size_t picp_help_words(int n_problems, int n_cu) { return (size_t)picp_help_rows(n_problems, n_cu) * 32 + (size_t)n_problems * 16; }
#if defined(__HIPCC__)
// What the gather does for correspondence i of problem p: (measurement m, world point w) -> packed x, y, z, u, v; an index
// outside its array leaves the marker and is counted (picp_batch_pack_kernel and the join's writing pass share this).
// in two halves, so that a caller can request the point and the pixel before it knows the slot they go to
struct PackedItem { float x, y, z, u, v; };
__device__ __forceinline__ PackedItem batch_pack_load(const BatchArgs& a, int p, const Pose& Xw, int m, int w) {
  const float* world = a.world + 3 * (size_t)p * a.world_stride;
  const float* meas = a.meas + 2 * (size_t)p * a.meas_stride;
  PackedItem it{__int_as_float((int)VO_DROPPED_BITS), 0.f, 0.f, 0.f, 0.f};
  if (m >= 0 && m < a.n_meas && w >= 0 && w < a.n_world) {
    it.x = world[3 * (size_t)w]; it.y = world[3 * (size_t)w + 1]; it.z = world[3 * (size_t)w + 2];
    if (a.X_world) { const float px = it.x, py = it.y, pz = it.z; pose_apply(Xw, px, py, pz, it.x, it.y, it.z); }   // PointCloud.h:80, as transform_batch_kernel
    it.u = meas[2 * (size_t)m]; it.v = meas[2 * (size_t)m + 1];
  } else if (a.n_bad) {
    atomicAdd(&a.n_bad[p], 1);        // dropped (marker) and counted: reported in stats_out[4p + 3]
  }
  return it;
}
__device__ __forceinline__ void batch_pack_store(const BatchArgs& a, int p, size_t i, const PackedItem& it) {
  float* dst = a.packed + (size_t)p * 5 * a.cap;
  dst[i] = it.x; dst[a.cap + i] = it.y; dst[2 * a.cap + i] = it.z; dst[3 * a.cap + i] = it.u; dst[4 * a.cap + i] = it.v;
}
__device__ __forceinline__ void batch_pack_item(const BatchArgs& a, int p, const Pose& Xw, size_t i, int m, int w) {
  batch_pack_store(a, p, i, batch_pack_load(a, p, Xw, m, w));
}
__device__ __forceinline__ Pose batch_pack_pose(const BatchArgs& a, int p) {
  Pose Xw;
#pragma unroll
  for (int k = 0; k < 9; ++k) Xw.R[k] = (k % 4 == 0) ? 1.f : 0.f;
  Xw.t[0] = Xw.t[1] = Xw.t[2] = 0.f;
  if (a.X_world) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = a.X_world[16 * (size_t)p + k];
    Xw = pose_from_T16(t);
  }
  return Xw;
}
#endif
// when is the launch-per-round form the faster one?  (measured: DESIGN.md section 4.1)
bool picp_batch_prefers_rounds(int n_problems, size_t cap, int n_iters, int n_cu);

// ---- frame-major grids ------------------------------------------------------------------------------
// A batched kernel runs `nb` workgroups for each of `n_frames` frames.  Launched as grid (nb, n_frames) consecutive
// workgroups of one frame are dealt round-robin over the 8 XCDs, so the frame's gather targets (its points, appearances,
// index tables: 0.2 .. 2 MB) are pulled into every XCD's L2.  From 8 frames on the launch is 1-D instead
// (frame_grid) and frame_block() maps blockIdx.x so that all workgroups of a frame share blockIdx.x mod 8, i.e. one XCD
// and its L2 (observed placement; speed only, nothing depends on it).
struct FrameBlock { int f, b, nb; bool live; };
#if defined(__HIPCC__)
__device__ __forceinline__ FrameBlock frame_block(int nb, int n_frames) {
  FrameBlock r;
  if (n_frames < 8) { r.f = (int)blockIdx.y; r.b = (int)blockIdx.x; r.nb = nb; r.live = true; return r; }
  const unsigned L = blockIdx.x, s = L >> 3;
  r.f = (int)(s / (unsigned)nb) * 8 + (int)(L & 7u);
  r.b = (int)(s % (unsigned)nb);
  r.nb = nb;
  r.live = r.f < n_frames;
  return r;
}
#endif
inline dim3 frame_grid(int nb, int n_frames) {
  return n_frames < 8 ? dim3((unsigned)nb, (unsigned)n_frames) : dim3(8u * (unsigned)((n_frames + 7) / 8) * (unsigned)nb);
}

// ---- geometry / matcher / join (geom.hip, match.hip) -------------------------
struct Workspace;  // scratch owned by the context

// d_scratch: compaction_scratch_ints(n) ints (per-workgroup counts / offsets)
size_t compaction_scratch_ints(int n);
size_t join_scratch_bytes(int n_img, int n_frames);      // d_scratch of launch_join[_batch]
size_t triangulate_scratch_bytes(int n, int n_frames);   // d_scratch of launch_triangulate[_batch]
hipError_t launch_project_points(hipStream_t st, const CamK& cam, const Pose& T, const float* d_world,
                                 int n, int keep_indices, float* d_out_uv, int* d_counts,
                                 int* d_scratch);

hipError_t launch_transform_points(hipStream_t st, const Pose& T, const float* d_in, int n,
                                   const int* d_n, float* d_out);
hipError_t launch_transform_points_devpose(hipStream_t st, const float* d_T16, const float* d_in, int n,
                                           const int* d_n, float* d_out);

hipError_t launch_triangulate(hipStream_t st, const float K[9], const Pose* X_host,
                              const float* d_X16, const int32_t* d_pairs, int n, const int* d_n,
                              const float* d_p1, int n1, const float* d_p2, int n2,
                              const float* d_app2, float* d_out_xyz, int32_t* d_out_pairs,
                              float* d_out_app, int* d_n_out, int* d_scratch);

hipError_t launch_join(hipStream_t st, const int32_t* d_img, int n_img, const int* d_n_img,
                       const int32_t* d_world, int n_world, const int* d_n_world, int n_ref,
                       int32_t* d_out, int* d_n_out, unsigned long long* d_table /* n_ref words */,
                       int* d_scratch);

// matcher variants: 1 = full scan (no workspace), 2 = bucket-pruned scan, 3 = cell-hash search, 4 / 5 = the
// exact-duplicate pass first ("hash-first", match.hip), then variant 2 / 3 for the queries it left open; the
// workspace of variant v holds match_workspace_bytes(v, nt, nq, n_frames) bytes
constexpr int MATCH_VARIANT_AUTO = 0x100;     // or-ed into a variant that the automatic rule picked (launch_match[_batch]): the
                                              //   exact-duplicate pass then asks for most of a frame's sample queries to have a copy
// after a call that ran variant 4 / 5 on d_prune_ws: *d_out = 1 when at least one frame took the exact-duplicate pass
hipError_t launch_match_hint(hipStream_t st, const void* d_prune_ws, int n_frames, int* d_out);
// after a call that ran WITHOUT the pass: *d_out = 1 when one of eight sample queries of some frame found its match at distance 0
hipError_t launch_match_hint_from_best(hipStream_t st, const unsigned long long* d_best, size_t best_stride, int nq_cap,
                                       const int* d_n1, const int* d_n2, int n_frames, int* d_out);
size_t match_pruned_workspace_bytes(int nt, int nq, int n_frames);
size_t match_cells_workspace_bytes(int nt, int nq, int n_frames);
size_t match_hash_workspace_bytes(int nt, int n_frames);
bool match_cells_supported(int nt, int nq);   // set sizes the cell-hash search takes (beyond: the bucket-pruned scan)
bool match_hash_supported(int nt, int n_frames);   // tree sizes the exact-duplicate pass takes (beyond: the general search alone)
inline size_t match_workspace_bytes(int variant, int nt, int nq, int n_frames) {
  const size_t hash = variant >= 4 ? match_hash_workspace_bytes(nt, n_frames) : 0;
  const int v = variant >= 4 ? variant - 2 : variant;
  return hash + (v == 3 ? match_cells_workspace_bytes(nt, nq, n_frames)
               : v == 2 ? match_pruned_workspace_bytes(nt, nq, n_frames) : 0);
}
// n_frames frames of identical set sizes, frame f at base + f*stride (strides in floats / pairs);
// d_best: n_frames*min(n1,n2) keys; d_scratch: n_frames * compaction_scratch_ints(min(n1,n2)) ints; d_n_out[n_frames]
// d_n1 / d_n2 (both or neither): ragged frames -- frame f holds d_n1[f] <= n1 and d_n2[f] <= n2 points (n1, n2 are then the
// capacities, the strides must be >= them) and picks its own tree (its larger set); always the full scan
hipError_t launch_match_batch(hipStream_t st, const float* d_a1, int n1, size_t a1_stride, const float* d_a2, int n2,
                              size_t a2_stride, float radius, int32_t* d_out_pairs, size_t out_stride, int* d_n_out,
                              unsigned long long* d_best, int* d_scratch, int n_cu, void* d_prune_ws, int n_frames,
                              int variant, const int* d_n1 = nullptr, const int* d_n2 = nullptr);
hipError_t launch_transform_batch(hipStream_t st, const float* d_T16, const float* d_in, int n, size_t stride,
                                  float* d_out, int n_frames);
hipError_t launch_triangulate_batch(hipStream_t st, const float K[9], const Pose* X_host, const float* d_X16,
                                    const int32_t* d_pairs, int n, const int* d_n, const float* d_p1, int n1,
                                    const float* d_p2, int n2, const float* d_app2, float* d_out_xyz,
                                    int32_t* d_out_pairs, float* d_out_app, int* d_n_out, int* d_scratch, int n_frames,
                                    size_t pairs_stride, size_t p1_stride, size_t p2_stride, size_t out_stride);
// sink (or null): the batched solver's arguments -- the writing pass then also gathers every pair it emits into the solver's
// packed arrays (what picp_batch_pack_kernel would do from the pairs it has just written); the caller sets sink->prepacked.
// join_fuses_gather(): whether launch_join_batch will honour a sink at these sizes (the one-launch form of small frames does not).
bool join_fuses_gather(int n_img, int n_world, int n_ref);
hipError_t launch_join_batch(hipStream_t st, const int32_t* d_img, int n_img, const int* d_n_img,
                             const int32_t* d_world, int n_world, const int* d_n_world, int n_ref, int32_t* d_out,
                             int* d_n_out, unsigned long long* d_table, int* d_scratch, int n_frames, size_t img_stride,
                             size_t world_stride, size_t out_stride, const BatchArgs* sink = nullptr);

// in-place exclusive scan of nb ints per frame (one workgroup per frame), total to total[frame] (and total2[frame])
hipError_t launch_scan(hipStream_t st, int* counts, int nb, int* total, int* total2 = nullptr, int n_frames = 1,
                       size_t counts_stride = 0);
// TreeNode_::fullSearch for every query (workspace: match_cells_workspace_bytes(nt, nq, 1))
hipError_t launch_radius_search(hipStream_t st, const float* d_tree, int nt, const float* d_qry, int nq, float radius,
                                int* d_offsets, int32_t* d_indices, int capacity, void* ws);

hipError_t launch_match(hipStream_t st, const float* d_a1, int n1, const float* d_a2, int n2,
                        float radius, int32_t* d_out_pairs, int* d_n_out,
                        unsigned long long* d_best /* min(n1,n2) u64 */, int* d_scratch, int n_cu,
                        void* d_prune_ws, int variant);

// ---- the map on the device (map.hip) ------------------------------------------------------------------
struct MapDev {
  float* pts = nullptr;                 // [cap][3]
  float* app = nullptr;                 // [cap][10]
  unsigned long long* table = nullptr;  // [tcap] open addressing, (tag << 32) | entry
  int* last = nullptr;                  // [tcap]
  int* hdr = nullptr;                   // [0] size [1] base of the last update [2] its rows [3] rows dropped for lack of room (never, unless misused)
  unsigned tcap = 0;                    // a power of two, >= 2 * cap
  int cap = 0;
};
size_t map_scratch_ints(int n_max);
// PointCloudVector<3>::update(T * cloud) (PointCloud.h:52-66,77-82); d_T16 / d_n may be null; d_scratch: map_scratch_ints(n_max) ints
hipError_t launch_map_update(hipStream_t st, const MapDev& m, const float* d_xyz, const float* d_app, int n_max, const int* d_n,
                             const float* d_T16, int* d_scratch);
hipError_t launch_map_rehash(hipStream_t st, const MapDev& m, int size_bound);      // empties the table, enters entries 0 .. size-1
hipError_t launch_map_history(hipStream_t st, float* d_hist16, const float* d_X16, int reset);
hipError_t launch_map_transform(hipStream_t st, const MapDev& m, const Pose& T, int size_bound);

// ---- epipolar initialisation, device half (epi.hip) ------------------------------------------------------
size_t epi_workspace_bytes();
// maxima of both point sets + the 45 sums of A^T A + the number of rows / of pairs with a bad index, into ws:
// [0,16) four float maxima  [16,32) int info[0] = rows used, info[1] = bad pairs  [32,48) vote counts  [64,424) 45 doubles
hipError_t launch_epi_front(hipStream_t st, const int32_t* d_pairs, int n_max, const int* d_n, const float* d_p1, int n1,
                            const float* d_p2, int n2, void* ws);
hipError_t launch_epi_vote(hipStream_t st, const float K[9], const Pose X[4], const int32_t* d_pairs, int n_max, const int* d_n,
                           const float* d_p1, int n1, const float* d_p2, int n2, void* ws);

}  // namespace vo
