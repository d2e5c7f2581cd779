/*
 * vo_hip.h -- C ABI of libvo_hip.so: the MI355X (gfx950) implementation of the
 * projective-ICP hot path of lucanunz/Visual-odometry.
 *
 * Every entry point replaces one interface of the reference (cited as
 * <file>:<line> under the reference tree).  The reference has no FFI: its
 * "plugin API" for this path is two C++ classes (Camera, PICPSolver) and a few
 * free functions; include/vo/ *.hpp re-create those on top of this ABI.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++ / torch types.
 *  - matrices are COLUMN-major, exactly the memory of the Eigen objects the
 *    reference passes around (defs.h:7-29): K = Eigen::Matrix3f (9 floats),
 *    T/X = Eigen::Isometry3f (4x4, 16 floats).
 *  - point arrays are the contiguous storage of the reference's std::vectors:
 *    Vector3fVector -> float[3n], Vector2fVector -> float[2n],
 *    Vector10fVector -> float[10n], IntPairVector -> int32_t[2n] (first,second).
 *  - functions return 0 (VO_OK) or a negative vo_status; vo_last_error() gives
 *    the message of the last failure on the calling thread.
 *  - unless the name ends in _dev, array arguments are HOST pointers: inputs
 *    are copied to the GPU, outputs copied back, and the call returns when the
 *    outputs are valid.  *_dev variants take DEVICE pointers (memory from
 *    vo_dev_alloc or any hipMalloc), enqueue on the context's stream and do not
 *    synchronise; counts are then produced in device memory.  Device arrays
 *    must start on an 8-byte boundary (hipMalloc gives 256; a sub-array at an
 *    even element offset keeps it): rows and index pairs move as 8-byte pieces.
 *  - a vo_ctx and the handles created from it must be used by one host thread
 *    at a time (the reference objects are not thread-safe either).
 *  - there is NO CPU fallback: every entry point fails with VO_ERR_NO_DEVICE if
 *    no gfx950 device can be used.
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_HIP_ABI_VERSION 1

typedef enum vo_status {
  VO_OK = 0,
  VO_ERR_INVALID_ARG = -1,
  VO_ERR_NO_DEVICE = -2,
  VO_ERR_HIP = -3,
  VO_ERR_OUT_OF_MEMORY = -4,
  VO_ERR_BAD_INDEX = -5,   /* a correspondence index is outside its array */
  VO_ERR_NOT_READY = -6    /* e.g. one_round before set_points */
} vo_status;

typedef struct vo_ctx vo_ctx;     /* one per (device, stream) */
typedef struct vo_picp vo_picp;   /* device twin of a PICPSolver */

int vo_abi_version(void);
const char *vo_last_error(void);

/* ---- context ----------------------------------------------------------- */
/* stream: a hipStream_t to enqueue on (e.g. torch's current stream), or NULL
 * to let the context create its own non-blocking stream. */
int vo_ctx_create(int device, void *stream, vo_ctx **out);
/* Handles made on a context (vo_picp, vo_graph, vo_kdtree) should be destroyed before it.  If one outlives its
 * context anyway, every use of it fails with VO_ERR_INVALID_ARG and its destroy still releases what it owns;
 * destroying a context twice is refused. */
int vo_ctx_destroy(vo_ctx *ctx);
int vo_ctx_synchronize(vo_ctx *ctx);
void *vo_ctx_stream(vo_ctx *ctx);
int vo_ctx_device(vo_ctx *ctx);
/* name of the device ("gfx950...") and number of compute units */
int vo_ctx_device_info(vo_ctx *ctx, char *name, int name_len, int *n_cu);

/* ---- hipGraph capture of a sequence of *_dev calls -------------------------------- */
/* Everything enqueued on the context's stream between begin and end (only *_dev entry
 * points: no host copies, no synchronisation, and every buffer the sequence needs must
 * already have been sized by a previous identical call) becomes one replayable graph:
 * a whole frame (match, join, transform, n rounds, triangulate = ~80 launches) then
 * costs one launch on the host.  Device pointers and counts are baked in; data and
 * device-side counts may change between replays.  A graph (like a vo_picp or a vo_event)
 * must be destroyed before the context it was made on.  Between begin and end every entry
 * point that copies host memory, allocates or waits (the forms without _dev, vo_ctx_synchronize,
 * vo_dev_alloc ...) and every *_dev call that would have to grow a workspace is refused with
 * VO_ERR_NOT_READY / VO_ERR_HIP before it touches the stream: the capture stays valid. */
typedef struct vo_graph vo_graph;
int vo_ctx_begin_capture(vo_ctx *ctx);
int vo_ctx_end_capture(vo_ctx *ctx, vo_graph **out);
int vo_graph_launch(vo_graph *g);          /* on the stream of the context it was captured on */
int vo_graph_destroy(vo_graph *g);

/* Ordering between contexts (= streams) of one device.  Work enqueued on one context can be made to
 * wait for a point in another context's stream without blocking the host: e.g. uploads or the
 * matcher of frame t+1 (it depends on the appearances alone) on a second context while frame t runs on
 * the first (pipeline.py: SequencePipeline(overlap_match=True); measured in DESIGN.md section 5). */
typedef struct vo_event vo_event;
int vo_event_create(vo_ctx *ctx, vo_event **out);
int vo_event_record(vo_event *ev, vo_ctx *ctx);      /* marks the current end of ctx's stream */
int vo_ctx_wait_event(vo_ctx *ctx, vo_event *ev);    /* later work on ctx waits for the marked point */
int vo_event_destroy(vo_event *ev);

/* device memory helpers for callers without a HIP runtime of their own */
int vo_dev_alloc(vo_ctx *ctx, size_t bytes, void **dptr);
int vo_dev_free(vo_ctx *ctx, void *dptr);
int vo_memcpy_h2d(vo_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes); /* synchronous */
int vo_memcpy_d2h(vo_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes); /* synchronous */

/* ---- Camera::projectPoints (camera.cpp:16-37; projectPoint camera.h:25-37) */
/* out_uv has room for n points.  keep_indices!=0: *n_out = n and invalid
 * points are (-1,-1); keep_indices==0: valid points only, input order kept.
 * *n_inside = the reference's return value (number of points inside). */
int vo_project_points(vo_ctx *ctx, int rows, int cols, int z_near, int z_far, const float K[9],
                      const float T[16], const float *world_xyz, int n, int keep_indices,
                      float *out_uv, int *n_out, int *n_inside);
/* device form: d_counts[0] = n_out, d_counts[1] = n_inside */
int vo_project_points_dev(vo_ctx *ctx, int rows, int cols, int z_near, int z_far, const float K[9],
                          const float T[16], const float *d_world_xyz, int n, int keep_indices,
                          float *d_out_uv, int *d_counts);

/* ---- PICPSolver (picp_solver.h:18-79, picp_solver.cpp) ----------------- */
/* ctor: threshold 1000, damping 1, min inliers 0 (picp_solver.cpp:6-14).  Damping has no setter -- as in the reference -- and the
 * default mode's unpivoted 6x6 solve relies on it: H = sum(lambda J^T J) + 1 * I is positive definite whatever the frame holds. */
int vo_picp_create(vo_ctx *ctx, vo_picp **out);
int vo_picp_destroy(vo_picp *s);
/* init(camera, world, image) (picp_solver.cpp:16-23).  The reference stores
 * raw pointers to the caller's vectors; this copies them to the GPU. */
int vo_picp_set_camera(vo_picp *s, int rows, int cols, int z_near, int z_far, const float K[9],
                       const float T[16]);
int vo_picp_set_points(vo_picp *s, const float *world_xyz, int n_world, const float *meas_uv,
                       int n_meas);
/* borrowed device arrays; must stay valid until the next set_points* call */
int vo_picp_set_points_dev(vo_picp *s, const float *d_world_xyz, int n_world,
                           const float *d_meas_uv, int n_meas);
int vo_picp_set_pose(vo_picp *s, const float T[16]);            /* camera.h:50 */
/* same from a 4x4 in device memory: takes effect at the next one_round/solve call, on
 * the stream, with no host sync (d_T16 must stay valid until then) */
int vo_picp_set_pose_dev(vo_picp *s, const float *d_T16);
int vo_picp_set_kernel_threshold(vo_picp *s, float thr);        /* picp_solver.h:35 */
int vo_picp_get_kernel_threshold(vo_picp *s, float *thr);       /* picp_solver.h:33 */
/* oneRound(correspondences, keep_outliers) (picp_solver.cpp:98-112):
 * pairs = (measurement index, world index).  Enqueues one Gauss-Newton
 * iteration -- ONE kernel launch -- and returns without waiting; the
 * pose/statistics getters are the synchronisation points, and the first of
 * them after a run of calls enqueues the launch that finishes the last round
 * (its 6x6 solve, H, b, statistics): a loop of oneRound calls followed by
 * camera(), as vo_complete.cpp:163-168, costs one launch per call plus one.
 * Like the reference (picp_solver.cpp:62) every call honours the array it is
 * given: the pairs are compared IN FULL (memcmp) with the host copy of what is
 * on the GPU and uploaded again when anything differs, so editing the vector in
 * place between two rounds is seen.  When the array has the length of the
 * packed one the round is enqueued first and the comparison runs while the GPU
 * works; a difference then repeats that round on the new pairs (it had written
 * nothing a repeat does not overwrite).  Once two calls in a row have matched --
 * the reference's loop -- a call enqueues its round and up to seven more as ONE
 * graph launch (captured the third time a launch geometry asks for it; the last
 * window of a run is cut to the remembered length of the previous run), and the
 * calls that follow only compare and claim theirs; rounds
 * that ran ahead and are not claimed (the loop ended, the pairs / points /
 * parameters changed) are ignored or repeated, never seen: every result is the
 * one a closed solve of the counted rounds gives.  VO_PICP_RUN_AHEAD=0..14 in the
 * environment (read by vo_picp_create) bounds the look-ahead; at most that many
 * rounds of GPU time are spent for nothing per loop.  As in the reference it cannot fail on
 * "too few inliers" (min_num_inliers is 0 with no setter). */
int vo_picp_one_round(vo_picp *s, const int32_t *pairs, int n_pairs, int keep_outliers);
/* Bookkeeping of the above (any pointer may be NULL): rounds enqueued whose finishing launch is still to come, calls
 * enqueued ahead of their comparison since the handle was made, and how many of those had to be repeated. */
int vo_picp_chain_info(vo_picp *s, int *open_rounds, unsigned long long *speculative, unsigned long long *repeated);
/* n_iters x oneRound with no host round trip in between */
int vo_picp_solve(vo_picp *s, const int32_t *pairs, int n_pairs, int keep_outliers, int n_iters);
/* Explicit form of the same, for callers that iterate on one fixed set: hand the pairs over once
 * (always uploaded), then run rounds on them with no per-call comparison at all. */
int vo_picp_set_correspondences(vo_picp *s, const int32_t *pairs, int n_pairs);
int vo_picp_rounds(vo_picp *s, int keep_outliers, int n_iters);
/* Launch-graph bookkeeping of a handle (any pointer may be NULL): whether multi-round solves are replayed from a captured
 * hipGraph (1) or issued as plain launches (0: VO_PICP_GRAPH=0, or a capture failed), how many graphs are cached, and how
 * many captures failed.  A failed capture does not fail the solve -- the same kernels run as plain launches -- but it is
 * reported once through vo_last_error() and counted here instead of passing unnoticed. */
int vo_picp_graph_info(vo_picp *s, int *use_graph, int *n_graphs, int *n_failures);
/* Reference-order arithmetic (off by default).  on != 0: every later round of this handle is computed
 * with the reference's own rounding -- per-correspondence terms unfused, (J0r*J0c + J1r*J1c)*lambda,
 * H / b / chi summed sequentially in correspondence order (picp_solver.cpp:62-95), Eigen's pivoted
 * LDLT with true divisions (:109), sin/cos in double rounded to float (utils.h:16-78) -- so pose,
 * H, b, chi and the inlier count are BIT-IDENTICAL to the reference's scalar float32 arithmetic
 * (as restated by oracle/: tests/test_gpu_exact.py).  One workgroup, all rounds in one launch:
 * a few microseconds per round at the <= 127 points per frame of the reference's dataset, 0.18 ms
 * per round at 50k (the serial chain of 50 000 dependent float adds).  The default (fast) mode differs from it by rounding only (tree reduction,
 * one FMA per product, Newton-refined hardware reciprocals, unpivoted LDLT, float sincos): a gate or chi^2 decision can differ
 * from the reference's only for a correspondence whose reference-order value lies within a few ulp of the gate -- measured on
 * correspondences planted at every gate (tests/test_gpu_gates.py, same pose in): <= 2 ulp for the depth gates, <= 3.7 for the
 * image gates, <= 276 ulp of the threshold for chi^2 (3e-5 relative); the test holds 4 / 8 / 1024. */
int vo_picp_set_exact(vo_picp *s, int on);
/* device pairs; d_n_pairs (may be NULL) points at a device int that overrides
 * n_pairs (<= n_pairs), so the output of the join kernel can be consumed
 * without a host round trip. */
int vo_picp_solve_dev(vo_picp *s, const int32_t *d_pairs, int n_pairs, const int *d_n_pairs,
                      int keep_outliers, int n_iters);
int vo_picp_get_pose(vo_picp *s, float T[16]);                  /* camera(), picp_solver.h:41 */
int vo_picp_get_pose_dev(vo_picp *s, float *d_T16);             /* async copy on the stream */
/* device address of the solver's own 4x4 pose (column-major), valid for the life of the
 * handle and rewritten by every solve: lets a consumer kernel read the result in place
 * (rounds of vo_picp_one_round reach it with the next getter call -- this one included) */
int vo_picp_pose_dev_ptr(vo_picp *s, const float **d_T16);
int vo_picp_get_stats(vo_picp *s, float *chi_inliers, float *chi_outliers, int *num_inliers); /* :44-50 */
/* H (6x6 col-major, damping included, as _H after oneRound) and b of the last round */
int vo_picp_get_system(vo_picp *s, float H[36], float b[6]);

/* Batched solver: n_problems independent (camera, points, pairs) problems,
 * every one iterated n_iters times inside one launch.  Arrays are DEVICE
 * pointers; problem p uses world[p*world_stride..], meas[p*meas_stride..],
 * pairs[p*pairs_stride..] (strides in elements of the respective type:
 * points, points, pairs) and n_pairs[p] pairs.  All share rows/cols/z/K/thr.
 * d_T0: n_problems initial poses (16 floats each) or NULL for identity;
 * d_T_out: n_problems final poses; d_stats_out (may be NULL): per problem
 * {chi_inliers, chi_outliers, (float)num_inliers, (float)n_bad} -- n_bad = pairs of the problem whose
 * index lies outside its point arrays: they are dropped (the single-problem entry points report the same
 * condition as VO_ERR_BAD_INDEX from their getters). */
/* Two forms, same results up to the summation order of H and b: one workgroup per problem with all rounds
 * inside one launch (many problems: streaming bound), or one launch per round with many workgroups per
 * problem (a few problems: the single-problem kernels with the problem as a grid dimension; 2.5x faster at one
 * problem of 50k, equal at ~10).  form 0 (default) picks by a cost model, 1 / 2 force one.
 * With one workgroup per problem and at most 0.65 problems per CU (and >= 18 432 correspondences of capacity per problem)
 * the launch has one workgroup per CU and the waves of those without a problem take work off the others' every round
 * (csrc/picp.hip, picp_batch_shared_kernel: 1.9x at 32 problems of 50k, 1.1x at 128; beyond, the problems' own
 * workgroups already draw what the memory side delivers).  The result does not depend on when, or whether, a helper
 * wave runs; problems with the same data in one call get the same bits.  VO_PICP_SHARE=0 in the environment turns it off.
 * vo_picp_batch_info: what the context's LAST batched call ran as -- 1 one launch per round, 2 one workgroup per problem,
 * 3 reference-order arithmetic, 4 one workgroup per problem with helpers (0: no call yet) -- and its workgroups per
 * launch.  Either pointer may be NULL. */
int vo_picp_batch_set_form(vo_ctx *ctx, int form);
int vo_picp_batch_info(vo_ctx *ctx, int *form, int *workgroups);
int vo_picp_solve_batch_dev(vo_ctx *ctx, int n_problems, int rows, int cols, int z_near, int z_far,
                            const float K[9], float kernel_threshold, int keep_outliers,
                            const float *d_world_xyz, size_t world_stride, const float *d_meas_uv,
                            size_t meas_stride, const int32_t *d_pairs, size_t pairs_stride,
                            const int *d_n_pairs, const float *d_T0, int n_iters, float *d_T_out,
                            float *d_stats_out);

/* ---- compute_correspondences_images (vo_complete.cpp:12-49) ------------ */
/* Exact nearest neighbour within `radius` in the 10-D appearance space,
 * replacing TreeNode_::bestMatchFull (eigen_kdtree.h:90-115): the larger set
 * is searched (ties: a1), the smaller set queries in ascending index, a hit
 * needs squared distance < radius*radius (strict), pairs are emitted as
 * (index in a1, index in a2).  Exact-distance ties go to the lowest index.
 * out_pairs has room for min(n1,n2) pairs. */
/* All variants return identical pairs.  mode 0 (default): pick by size and frame count; 1: scan every
 * (query, tree point) pair; 2: counting-sort both sets into 1024 buckets along the two appearance
 * components of largest spread and scan only the buckets within the radius (LDS-tiled); 3: counting-sort
 * both sets into a 4-D grid of cells no narrower than the radius and visit, per query, the <= 81 cells
 * around it (about 25 candidates per query on uniform appearances); 4 / 5: first answer every query that has a
 * bitwise copy in the tree (appearances are copied from frame to frame; a copy is at distance 0, the minimum) through
 * hash tables cut by hash, then run 2 / 3 for the remaining queries only -- what mode 0 does from 8 frames per call on
 * at the sizes where it sorts (sets of up to 204 800 points; beyond, 2 / 3 alone).  In mode 5 a frame that is left with
 * FEW open queries (at most 1/16 of its queries, and 2560: the new landmarks of a tracking frame) does not sort its tree at
 * all: the open queries are ordered by cell and the tree is streamed once past them.  Matcher stage of 200 x 50k frames:
 * 0.52 ms when every query has a copy, 0.73 at 1 % open, 0.80 at 5 %, 1.27 at 25 % (profiles/r05_bench_line.json); mode 3
 * alone 1.10; data WITHOUT any copies is noticed from eight sampled queries per frame and skips the tables and the lookup
 * (1.29 ms) -- and, in mode 0, the NEXT calls leave the pass out altogether (1.10 ms): what the previous call found travels to
 * the host behind the stream, "no frame took the pass" twice in a row switches it off for 16 calls, a sample query that the search finds at
 * distance 0 switches it on again at once, vo_match_set_mode() forgets what was learnt.  Partly copied data (10 - 90 % of
 * the queries without a copy) pays 0.05 - 0.2 ms for the pass -- ask for mode 3 there. */
int vo_match_set_mode(vo_ctx *ctx, int mode);
int vo_match_appearances(vo_ctx *ctx, const float *a1, int n1, const float *a2, int n2,
                         float radius, int32_t *out_pairs, int *n_out);
int vo_match_appearances_dev(vo_ctx *ctx, const float *d_a1, int n1, const float *d_a2, int n2,
                             float radius, int32_t *d_out_pairs, int *d_n_out);

/* ---- TreeNode_::fullSearch (eigen_kdtree.h:56-71) for a whole query set -- */
/* Every tree point within `radius` of each query (squared distance < radius*radius, strict, as
 * bruteForceSearch brute_force_search.h:3-20), exactly -- the same 4-D cell search as matcher mode 3.
 * CSR result: query i owns indices[offsets[i] .. offsets[i+1]) (tree indices; the order inside one query's
 * list is unspecified, as it is in the reference, where it follows the tree traversal).  `capacity` = room
 * in `indices`; *n_total = hits found.  If n_total > capacity the call fails with VO_ERR_INVALID_ARG and
 * offsets/n_total tell the caller how much room to bring.  Unlike the matcher the roles are explicit: the
 * first set is searched whatever the sizes.  The approximate modes of the reference (fastSearch,
 * bestMatchFast: descend one side of each PCA split) depend on its tree: vo_kdtree_* below; their
 * answers are subsets of this call's / of vo_match_appearances'.  Sets of up to 1 835 008 points each (1024 slices of the
 * level-1 sort); the device arrays on 8-byte boundaries like every device entry point. */
int vo_radius_search(vo_ctx *ctx, const float *tree_app, int n_tree, const float *query_app, int n_q,
                     float radius, int32_t *offsets, int32_t *indices, int capacity, int *n_total);
/* device form: d_offsets[n_q + 1]; d_offsets[n_q] = hits found (also when > capacity: the surplus is dropped) */
int vo_radius_search_dev(vo_ctx *ctx, const float *d_tree_app, int n_tree, const float *d_query_app,
                         int n_q, float radius, int32_t *d_offsets, int32_t *d_indices, int capacity);

/* ---- TreeNode_ in its approximate modes (eigen_kdtree.h:18-52,75-85) ------------------------ */
/* bestMatchFast and fastSearch descend ONE side of every PCA split plane and brute-force the leaf they reach: their
 * answers depend on the tree (split directions, order of the points inside a leaf), so the tree is a handle.
 * vo_kdtree_create builds it on the host like the TreeNode_ constructor (:18-38; mean/covariance in float in array
 * order, eigen_covariance.h:5-43 with a double Jacobi for the eigen-solver -- the split normal's sign is fixed by
 * convention (largest-magnitude component positive), which Eigen's SelfAdjointEigenSolver does not promise: every leaf
 * holds the same SET of points as the reference's up to the solver's sign / degeneracy, the order inside a leaf (and with
 * it bestMatchFast's first-minimum tie and fastSearch's result order) may differ --, the two-pointer partition of split.h:8-34,
 * recursion while a node holds >= max_points_in_leaf points; a node whose points all fall on one side becomes a leaf --
 * the reference recurses forever there) and uploads it; queries run on the GPU, one lane per query.  points: host
 * float[10n] (the reference's 11-vectors carry the index in slot 0; here the index is implicit).  Neither mode is used
 * by an executable of the reference; the exact modes are vo_match_appearances (bestMatchFull) and vo_radius_search
 * (fullSearch), whose answers do not depend on any tree. */
typedef struct vo_kdtree vo_kdtree;
int vo_kdtree_create(vo_ctx *ctx, const float *points_app, int n, int max_points_in_leaf, vo_kdtree **out);
int vo_kdtree_destroy(vo_kdtree *tree);       /* before the context it was made on */
int vo_kdtree_info(vo_kdtree *tree, int *n_points, int *n_nodes, int *n_leaves);
/* bestMatchFast (:75-85) for every query: out_index[i] = index (in points_app) of the closest point OF THE QUERY'S LEAF
 * with squared distance < radius*radius (strict; the first minimum in leaf order), or -1 */
int vo_kdtree_best_match_fast(vo_kdtree *tree, const float *query_app, int n_q, float radius, int32_t *out_index);
int vo_kdtree_best_match_fast_dev(vo_kdtree *tree, const float *d_query_app, int n_q, float radius, int32_t *d_out_index);
/* fastSearch (:40-52) for every query, CSR like vo_radius_search: query i owns indices[offsets[i] .. offsets[i+1]), the
 * points of its leaf within the radius IN LEAF ORDER (the order bruteForceSearch pushes them, brute_force_search.h:3-20);
 * n_total > capacity: VO_ERR_INVALID_ARG with offsets / n_total filled in */
int vo_kdtree_fast_search(vo_kdtree *tree, const float *query_app, int n_q, float radius, int32_t *offsets,
                          int32_t *indices, int capacity, int *n_total);
int vo_kdtree_fast_search_dev(vo_kdtree *tree, const float *d_query_app, int n_q, float radius, int32_t *d_offsets,
                              int32_t *d_indices, int capacity);

/* ---- extract_correspondences_world (vo_complete.cpp:52-66) ------------- */
/* For each image pair (ref,cur) in order, the FIRST world pair (ref',w) with
 * ref'==ref gives (cur,w); image pairs without partner are dropped.
 * out_pairs has room for n_img pairs. */
int vo_join_correspondences(vo_ctx *ctx, const int32_t *img_pairs, int n_img,
                            const int32_t *world_pairs, int n_world, int32_t *out_pairs,
                            int *n_out);
/* device form; the counts may come from device memory (NULL -> use n_img / n_world);
 * n_ref = size of the reference index space (all .first values < n_ref). */
int vo_join_correspondences_dev(vo_ctx *ctx, const int32_t *d_img_pairs, int n_img,
                                const int *d_n_img, const int32_t *d_world_pairs, int n_world,
                                const int *d_n_world, int n_ref, int32_t *d_out_pairs,
                                int *d_n_out);

/* ---- Isometry3f * point set (PointCloud.h:77-82, vo_daKnown.cpp:144) --- */
int vo_transform_points(vo_ctx *ctx, const float T[16], const float *in_xyz, int n,
                        float *out_xyz);
/* device form: the isometry comes from the host (T) or, when d_T16 is non-NULL, from device
 * memory (column-major 4x4, e.g. vo_picp_pose_dev_ptr of the previous frame's solve -- the
 * X_curr * triangulated_pc of vo_complete.cpp:159 with no host round trip); d_n (may be NULL)
 * points at a device count <= n. */
int vo_transform_points_dev(vo_ctx *ctx, const float T[16], const float *d_T16,
                            const float *d_in_xyz, int n, const int *d_n, float *d_out_xyz);

/* ---- triangulate_points (utils.cpp:51-134; triangulate_point :36-49) --- */
/* pairs = (index in p1, index in p2); X = pose of the first camera in the
 * frame of the second.  Survivors are written densely in input order:
 * out_xyz[k], out_pairs[k] = (index in p2, k) (may be NULL: overload v1),
 * out_app[k] = app2[index in p2] when app2/out_app are non-NULL (overload
 * v3).  Output arrays have room for n pairs.  Returns the count in *n_out. */
int vo_triangulate(vo_ctx *ctx, const float K[9], const float X[16], const int32_t *pairs, int n,
                   const float *p1_uv, int n1, const float *p2_uv, int n2, const float *app2,
                   float *out_xyz, int32_t *out_pairs, float *out_app, int *n_out);
int vo_triangulate_dev(vo_ctx *ctx, const float K[9], const float X[16], const float *d_X16,
                       const int32_t *d_pairs, int n, const int *d_n, const float *d_p1_uv, int n1,
                       const float *d_p2_uv, int n2, const float *d_app2, float *d_out_xyz,
                       int32_t *d_out_pairs, float *d_out_app, int *d_n_out);

/* ---- estimate_transform (epipolar_utils.cpp:176-213) --------------------- */
/* Relative pose of the first camera in the frame of the second from >= 8 image correspondences
 * pairs = (index in p1, index in p2): normalised 8-point fundamental (:103-144, normalisation over
 * ALL n1 / n2 points, :48-65), E = K^T F K, the four (R, +-t) candidates (:146-174) and the
 * cheirality vote -- how many correspondences triangulate under each candidate (the triangulation kernel's per-pair
 * arithmetic, counted) -- which keeps the first candidate with the most survivors.  Host linear algebra in double.
 * As in the reference t is read off R*E un-normalised (:163-164): |t| is the singular value
 * of E, which is what fixes the scale of a monocular sequence.  Fewer than 8 pairs:
 * VO_ERR_INVALID_ARG (the reference prints and exits, :105-108). */
int vo_estimate_transform(vo_ctx *ctx, const float K[9], const int32_t *pairs, int n,
                          const float *p1_uv, int n1, const float *p2_uv, int n2, float X_out[16]);
/* The same from arrays in device memory -- the matcher's pairs and the two images as they lie in HBM; *d_n_pairs (or
 * NULL) <= n_max pairs are live.  The sums over the correspondences run on the GPU (epi.hip): per-axis maxima of both
 * images (:50-56), the 45 distinct entries of A^T A in double (:114-126), and the cheirality vote of all four candidates in
 * one launch; the 9 x 9 eigen-solve and the decomposition of E stay on the host in double.  Two small read-backs;
 * X_out on the host.  (vo_estimate_transform is this after three uploads.) */
int vo_estimate_transform_dev(vo_ctx *ctx, const float K[9], const int32_t *d_pairs, int n_max, const int *d_n_pairs,
                              const float *d_p1_uv, int n1, const float *d_p2_uv, int n2, float X_out[16]);

/* ---- many independent frame pairs at once (throughput form of vo_complete.cpp:156-173) ---- */
/* For each of n_frames independent frame pairs: match -> join -> X_prev * model -> n_iters rounds
 * from the identity -> triangulate, every stage one batched launch (frame = a grid dimension) and
 * the solver the batched kernel of vo_picp_solve_batch_dev.  All frames share the set sizes, the
 * camera and the solver settings; per-frame counts are produced in device memory.  All pointers
 * are DEVICE pointers; frame f of an array lives at base + f * (items per frame).  Enqueues on
 * the context's stream and returns. */
typedef struct vo_frame_batch {
  int n_frames;
  int n_ref, n_cur;              /* points in every reference / current image */
  int n_model, n_model_pairs;    /* model points and (ref index, model index) pairs per frame */
  const float *ref_app, *cur_app;   /* [n_frames][n_ref|n_cur][10] */
  const float *ref_pts, *cur_pts;   /* [n_frames][n_ref|n_cur][2] */
  const float *model;               /* [n_frames][n_model][3] */
  const int32_t *model_pairs;       /* [n_frames][n_model_pairs][2] */
  const float *X_prev;              /* [n_frames][16] column-major, or NULL for identity */
  int rows, cols, z_near, z_far;
  float K[9];
  float kernel_threshold;
  int keep_outliers;
  int n_iters;
  float radius;                     /* appearance radius (0.1 in the reference) */
  /* outputs; q = min(n_ref, n_cur) items of room per frame */
  int32_t *matches;                 /* [n_frames][q][2]  (ref index, cur index) */
  int32_t *joined;                  /* [n_frames][q][2]  (cur index, model index) */
  float *model_moved;               /* [n_frames][n_model][3]  X_prev * model, or NULL: the moved cloud is not wanted -- the solver
                                     * then moves the points it gathers itself (same arithmetic, one pass and 12 B per model point less) */
  float *poses;                     /* [n_frames][16] */
  float *stats;                     /* [n_frames][4]: chi_inliers, chi_outliers, num_inliers, bad-index pairs dropped (may be NULL) */
  float *tri_xyz;                   /* [n_frames][q][3] */
  int32_t *tri_pairs;               /* [n_frames][q][2]  (cur index, slot) */
  float *tri_app;                   /* [n_frames][q][10] or NULL */
  int *counts;                      /* [3][n_frames]: matches, joined pairs, triangulated points */
} vo_frame_batch;
int vo_frames_batch_dev(vo_ctx *ctx, const vo_frame_batch *batch);
/* The same call for frames of DIFFERENT sizes -- the reference's own sequence has 14..127 points per frame
 * (vo_complete.cpp:150-157 runs the loop body on whatever each measurement file holds).  The counts of `batch` are then the
 * capacities (= the strides between frames); frame f really holds sizes->n_ref[f] <= n_ref and n_cur[f] <= n_cur points and
 * n_model_pairs[f] <= n_model_pairs model pairs (device int arrays of n_frames entries).  Every frame picks its own tree --
 * its larger set, the reference image on ties -- exactly like the single-frame call (vo_complete.cpp:15-20); the matcher
 * runs its full scan or, from the sizes on where sorting pays, the cell-hash search.  Outputs, counts and strides as above. */
typedef struct vo_frame_sizes {
  const int *n_ref, *n_cur, *n_model_pairs;
} vo_frame_sizes;
int vo_frames_batch_ragged_dev(vo_ctx *ctx, const vo_frame_batch *batch, const vo_frame_sizes *sizes);
/* The matcher alone for n_frames pairs of appearance sets: frame f = d_a1 + f*cap1*10 (d_n1[f] <= cap1 rows) against
 * d_a2 + f*cap2*10 (d_n2[f] <= cap2 rows); d_n1 = d_n2 = NULL: every frame holds cap1 / cap2 rows.  d_out_pairs: [n_frames][min(cap1, cap2)][2] (ref index, cur index), d_n_out[f] = pairs found.  Replaces
 * n_frames calls of compute_correspondences_images (vo_complete.cpp:12-49): the up-front matching of a whole sequence. */
int vo_match_appearances_batch_dev(vo_ctx *ctx, int n_frames, const float *d_a1, int cap1, const int *d_n1,
                                   const float *d_a2, int cap2, const int *d_n2, float radius, int32_t *d_out_pairs,
                                   int *d_n_out);

/* ---- the map (PointCloud.h:52-66; vo_complete.cpp:145-147,175-176,181-183) ---------------------------------------
 * PointCloudVector<3> `map` of vo_complete, resident on the GPU.  map.update(cloud) is the reference's upsert keyed by
 * EXACT equality of the ten appearance floats (operator==: -0 equals +0, a row with a NaN equals nothing, itself
 * included): for every point of the cloud, in order, the first entry with an equal appearance gets the point, otherwise
 * the pair is appended (and found by later points of the same cloud).  Equivalent, and what the kernels compute
 * (map.hip): every class of equal appearances keeps the appearance bits of its first occurrence ever and the point of
 * its last occurrence so far; new classes are appended in the order of their first occurrence in the cloud; NaN rows
 * are always appended.  O(cloud) per update through a hash table of first occurrences (the reference: O(cloud x map)).
 * All calls are enqueued on the context's stream; only vo_map_size / vo_map_read / vo_map_get_history wait.
 * The arrays grow by themselves (a stream synchronisation and a copy when they do: give vo_map_create the capacity a
 * run will need to avoid it; growing is refused inside a graph capture). */
typedef struct vo_map vo_map;
int vo_map_create(vo_ctx *ctx, int capacity, vo_map **out);
int vo_map_destroy(vo_map *m);
int vo_map_clear(vo_map *m);                                   /* empty map, history = identity */
/* map.update(T * cloud): d_xyz [n_max][3], d_app [n_max][10] (8-byte aligned), *d_n_rows (or NULL) <= n_max rows are
 * live, d_T16 (or NULL: identity) a column-major 4x4 in device memory applied to every point -- `history *
 * triangulated_pc` of vo_complete.cpp:175 with d_T16 = vo_map_history_dev_ptr() -- as PointCloud.h:77-82 does. */
int vo_map_update_dev(vo_map *m, const float *d_xyz, const float *d_app, int n_max, const int *d_n_rows, const float *d_T16);
int vo_map_update(vo_map *m, const float *xyz, const float *app, int n, const float T16[16] /* or NULL */);   /* host arrays */
/* the `history` isometry of vo_complete, kept on the device: reset = X^-1 (vo_complete.cpp:146), step = history * X^-1
 * (:176), both from a pose in device memory (e.g. vo_picp_pose_dev_ptr), in the reference's float arithmetic */
int vo_map_history_reset_dev(vo_map *m, const float *d_X16);
int vo_map_history_step_dev(vo_map *m, const float *d_X16);
int vo_map_history_dev_ptr(vo_map *m, const float **d_T16);
int vo_map_get_history(vo_map *m, float T16[16]);
int vo_map_transform(vo_map *m, const float T16[16]);           /* map = H * map (vo_complete.cpp:183), in place */
int vo_map_size(vo_map *m, int *n);
/* copies min(size, capacity) entries to the host (either array may be NULL); *n_out = size */
int vo_map_read(vo_map *m, float *xyz, float *app, int capacity, int *n_out);
/* the arrays in device memory (entries [0, *d_size)); they move when the map grows */
int vo_map_dev_ptrs(vo_map *m, const float **d_xyz, const float **d_app, const int **d_size);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
