// kdtree.hip -- the reference's PCA kd-tree in its APPROXIMATE modes (eigen_kdtree.h): TreeNode_::bestMatchFast
// (:75-85) and TreeNode_::fastSearch (:40-52) descend ONE side of every split plane and brute-force the leaf they
// reach, so their answers depend on the tree itself (split directions, point order inside a leaf).  Unused by every
// executable the reference builds; provided for API completeness.  The exact modes (bestMatchFull,
// fullSearch) are tree-independent and live in match.hip (vo_match_appearances, vo_radius_search).
//
// The tree is built on the host, once per point set, the way the TreeNode_ constructor does (:18-38): mean and
// covariance of the node's points accumulated in float in array order (eigen_covariance.h:5-30), the direction of
// largest variance (a cyclic Jacobi in double stands in for Eigen's SelfAdjointEigenSolver, :35-43; its sign is fixed
// by the convention in HostTree::direction -- same leaf sets as the reference, in-leaf order up to that sign), the in-place
// two-pointer partition of split.h:8-34 (which fixes the order of the points inside every leaf), recursion while a
// node holds >= max_points_in_leaf points.  Queries run on the GPU: one lane per query walks the <= ~log2(n) split
// planes (10-D dot products in the reference's left-to-right order, unfused) and scans its leaf.
// One guard is added: a node whose points all fall on one side of its plane (e.g. identical appearances) becomes a
// leaf -- the reference recurses forever there.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/vo_hip.h"
#include "vo_internal.h"

namespace vo {

constexpr int KD = 10;

struct KdNode {            // 96 bytes
  float mean[KD], normal[KD];
  int left, right;         // children (node indices), -1/-1 for a leaf
  int begin, end;          // the node's range in the tree-ordered point array
};

// symmetric eigen-decomposition by cyclic Jacobi rotations (double): a is destroyed (diagonal = eigenvalues),
// v receives the eigenvectors as columns
static void jacobi10(double a[KD][KD], double v[KD][KD]) {
  for (int i = 0; i < KD; ++i) for (int j = 0; j < KD; ++j) v[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < KD; ++p) for (int q = 0; q < KD; ++q) { if (p == q) diag += a[p][q] * a[p][q]; else off += a[p][q] * a[p][q]; }
    if (off <= 1e-26 * (diag + 1e-300)) break;
    for (int p = 0; p < KD - 1; ++p)
      for (int q = p + 1; q < KD; ++q) {
        const double apq = a[p][q];
        if (apq == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < KD; ++k) { const double x = a[k][p], y = a[k][q]; a[k][p] = c * x - s * y; a[k][q] = s * x + c * y; }
        for (int k = 0; k < KD; ++k) { const double x = a[p][k], y = a[q][k]; a[p][k] = c * x - s * y; a[q][k] = s * x + c * y; }
        for (int k = 0; k < KD; ++k) { const double x = v[k][p], y = v[k][q]; v[k][p] = c * x - s * y; v[k][q] = s * x + c * y; }
      }
  }
}

struct HostTree {
  std::vector<KdNode> nodes;
  std::vector<float> pts;      // n x 10 in tree order
  std::vector<int> index;      // original index of the point in slot i
  int max_leaf;

  float plane(const float* p, const KdNode& nd) const {          // (p - mean) . normal, left to right (eigen_kdtree.h:31)
    float s = 0.f;
    for (int i = 0; i < KD; ++i) s += (p[i] - nd.mean[i]) * nd.normal[i];
    return s;
  }
  void direction(int begin, int end, KdNode& nd) const {          // eigen_covariance.h:5-43
    float m[KD], cov[KD][KD];
    for (int i = 0; i < KD; ++i) { m[i] = 0.f; for (int j = 0; j < KD; ++j) cov[i][j] = 0.f; }
    const int k = end - begin;
    for (int it = begin; it < end; ++it) {
      const float* v = &pts[(size_t)it * KD];
      for (int i = 0; i < KD; ++i) m[i] += v[i];
      for (int i = 0; i < KD; ++i) for (int j = 0; j < KD; ++j) cov[i][j] += v[i] * v[j];
    }
    const float ik = (float)(1.0 / k);
    for (int i = 0; i < KD; ++i) m[i] *= ik;
    for (int i = 0; i < KD; ++i) for (int j = 0; j < KD; ++j) { cov[i][j] *= ik; }
    for (int i = 0; i < KD; ++i) for (int j = 0; j < KD; ++j) cov[i][j] -= m[i] * m[j];
    const float sc = (float)k / (float)(k - 1);
    double a[KD][KD], v[KD][KD];
    for (int i = 0; i < KD; ++i) for (int j = 0; j < KD; ++j) a[i][j] = (double)(cov[i][j] * sc);
    jacobi10(a, v);
    int best = 0;
    for (int i = 1; i < KD; ++i) if (a[i][i] > a[best][best]) best = i;
    // An eigenvector is defined up to its sign, and the sign decides which child is "left", hence the order of the points
    // inside a leaf.  Convention here (the CPU checker of the tests follows the same rule): the component of largest magnitude is positive (the
    // first one among equals).  Eigen's SelfAdjointEigenSolver has no such rule, so in-leaf order may differ from the
    // reference's; the leaf SETS do not depend on the sign.
    int big = 0;
    for (int i = 1; i < KD; ++i) if (fabs(v[i][best]) > fabs(v[big][best])) big = i;
    const double sgn = v[big][best] < 0.0 ? -1.0 : 1.0;
    for (int i = 0; i < KD; ++i) { nd.mean[i] = m[i]; nd.normal[i] = (float)(sgn * v[i][best]); }
  }
  int partition(int begin, int end, const KdNode& nd) {           // split.h:8-34
    int lower = begin, upper = end;
    float tmp[KD];
    while (lower != upper) {
      float* vl = &pts[(size_t)lower * KD];
      if (plane(vl, nd) < 0.f) {
        ++lower;
      } else {
        float* vu = &pts[(size_t)(upper - 1) * KD];
        memcpy(tmp, vl, sizeof(tmp)); memcpy(vl, vu, sizeof(tmp)); memcpy(vu, tmp, sizeof(tmp));
        const int ti = index[(size_t)lower]; index[(size_t)lower] = index[(size_t)upper - 1]; index[(size_t)upper - 1] = ti;
        --upper;
      }
    }
    return upper;
  }
  int build(int begin, int end) {                                 // eigen_kdtree.h:18-38 (pre-order node numbering)
    const int me = (int)nodes.size();
    nodes.push_back(KdNode{});
    nodes[(size_t)me].begin = begin; nodes[(size_t)me].end = end; nodes[(size_t)me].left = nodes[(size_t)me].right = -1;
    if (end - begin < max_leaf || end - begin < 2) return me;
    KdNode nd = nodes[(size_t)me];
    direction(begin, end, nd);
    const int middle = partition(begin, end, nd);
    if (middle == begin || middle == end) { nodes[(size_t)me] = nd; nodes[(size_t)me].left = nodes[(size_t)me].right = -1; return me; }
    nd.left = build(begin, middle);
    nd.right = build(middle, end);
    nodes[(size_t)me] = nd;
    return me;
  }
};

// ---- queries -----------------------------------------------------------------------------------
__device__ __forceinline__ void kd_leaf_of(const KdNode* __restrict__ nodes, const float q[KD], int& begin, int& end) {
  int node = 0;
  for (;;) {
    const KdNode& nd = nodes[node];
    if (nd.left < 0) { begin = nd.begin; end = nd.end; return; }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < KD; ++i) s += (q[i] - nd.mean[i]) * nd.normal[i];       // eigen_kdtree.h:47,80
    node = s < 0.f ? nd.left : nd.right;
  }
}

__device__ __forceinline__ float kd_sqdist(const float* __restrict__ p, const float q[KD]) {   // brute_force_search.h:14,34
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < KD; ++i) { const float d = p[i] - q[i]; s += d * d; }
  return s;
}

// bestMatchFast for every query: out[i] = original index of the closest leaf point with d2 < radius^2, or -1
__global__ __launch_bounds__(256) void kd_best_fast_kernel(const KdNode* __restrict__ nodes, const float* __restrict__ pts,
                                                           const int* __restrict__ index, const float* __restrict__ qry, int nq,
                                                           float radius, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  float q[KD];
#pragma unroll
  for (int k = 0; k < KD; ++k) q[k] = qry[(size_t)i * KD + k];
  int begin, end;
  kd_leaf_of(nodes, q, begin, end);
  float best = radius * radius;                                                  // brute_force_search.h:31
  int arg = -1;
  for (int it = begin; it < end; ++it) {
    const float d = kd_sqdist(pts + (size_t)it * KD, q);
    if (d < best) { best = d; arg = it; }                                        // :35 (strict: the first minimum in leaf order)
  }
  out[i] = arg >= 0 ? index[arg] : -1;
}

// fastSearch: WRITE false: counts[i] = number of leaf points with d2 < radius^2; WRITE true: their original indices, in
// leaf order, from offsets[i] on
template <bool WRITE>
__global__ __launch_bounds__(256) void kd_fast_search_kernel(const KdNode* __restrict__ nodes, const float* __restrict__ pts,
                                                             const int* __restrict__ index, const float* __restrict__ qry, int nq,
                                                             float radius, int* __restrict__ offsets, int32_t* __restrict__ indices,
                                                             int capacity) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  float q[KD];
#pragma unroll
  for (int k = 0; k < KD; ++k) q[k] = qry[(size_t)i * KD + k];
  int begin, end;
  kd_leaf_of(nodes, q, begin, end);
  const float sq = radius * radius;                                              // brute_force_search.h:10
  int n_hit = 0;
  const int at = WRITE ? offsets[i] : 0;
  for (int it = begin; it < end; ++it)
    if (kd_sqdist(pts + (size_t)it * KD, q) < sq) {                              // :14
      if (WRITE && at + n_hit < capacity) indices[at + n_hit] = index[it];
      ++n_hit;
    }
  if (!WRITE) offsets[i] = n_hit;
}

}  // namespace vo

using namespace vo;

struct vo_kdtree {
  vo_ctx* ctx = nullptr;
  unsigned long long ctx_id = 0;
  int n = 0, n_nodes = 0, n_leaves = 0, depth = 0;
  KdNode* d_nodes = nullptr;
  float* d_pts = nullptr;
  int* d_index = nullptr;
  void* d_q = nullptr; size_t q_cap = 0;        // staging of the host-pointer entry points
  void* d_o = nullptr; size_t o_cap = 0;
  void* d_i = nullptr; size_t i_cap = 0;
};

#define KD_CHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      (void)hipGetLastError();                                                                 \
      return vo_fail(_e == hipErrorOutOfMemory ? VO_ERR_OUT_OF_MEMORY : VO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                     hipGetErrorString(_e), __FILE__, __LINE__);                               \
    }                                                                                          \
  } while (0)

static int kd_grow(void** p, size_t* cap, size_t bytes, hipStream_t st) {
  if (bytes <= *cap) return VO_OK;
  if (*p) { KD_CHECK(hipStreamSynchronize(st)); (void)hipFree(*p); *p = nullptr; *cap = 0; }
  KD_CHECK(hipMalloc(p, bytes + bytes / 4 + 256));
  *cap = bytes + bytes / 4 + 256;
  return VO_OK;
}

extern "C" {

int vo_kdtree_create(vo_ctx* c, const float* app, int n, int max_points_in_leaf, vo_kdtree** out) {
  if (vo_ctx_capturing(c)) return vo_fail(VO_ERR_NOT_READY, "%s: not allowed inside a graph capture", __func__);
  if (!c || !out || n < 0 || (n > 0 && !app)) return vo_fail(VO_ERR_INVALID_ARG, "vo_kdtree_create: null argument or negative count");
  if (max_points_in_leaf < 1) return vo_fail(VO_ERR_INVALID_ARG, "vo_kdtree_create: max_points_in_leaf must be >= 1");
  *out = nullptr;
  KD_CHECK(hipSetDevice(vo_ctx_device(c)));
  hipStream_t st = reinterpret_cast<hipStream_t>(vo_ctx_stream(c));
  HostTree h;
  h.max_leaf = max_points_in_leaf;
  h.pts.assign(app, app + (size_t)n * KD);
  h.index.resize((size_t)n);
  for (int i = 0; i < n; ++i) h.index[(size_t)i] = i;
  h.build(0, n);
  vo_kdtree* t = new vo_kdtree();
  t->ctx = c; t->ctx_id = vo_ctx_id(c); t->n = n; t->n_nodes = (int)h.nodes.size();
  for (const KdNode& nd : h.nodes) if (nd.left < 0) ++t->n_leaves;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&t->d_nodes), sizeof(KdNode) * h.nodes.size());
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_pts), sizeof(float) * KD * (size_t)(n ? n : 1));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_index), sizeof(int) * (size_t)(n ? n : 1));
  if (e == hipSuccess) e = hipMemcpyAsync(t->d_nodes, h.nodes.data(), sizeof(KdNode) * h.nodes.size(), hipMemcpyHostToDevice, st);
  if (e == hipSuccess && n) e = hipMemcpyAsync(t->d_pts, h.pts.data(), sizeof(float) * KD * (size_t)n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess && n) e = hipMemcpyAsync(t->d_index, h.index.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);         // the sources are locals
  if (e != hipSuccess) {
    if (t->d_nodes) (void)hipFree(t->d_nodes);
    if (t->d_pts) (void)hipFree(t->d_pts);
    if (t->d_index) (void)hipFree(t->d_index);
    delete t;
    return vo_fail(e == hipErrorOutOfMemory ? VO_ERR_OUT_OF_MEMORY : VO_ERR_HIP, "vo_kdtree_create: %s", hipGetErrorString(e));
  }
  *out = t;
  return VO_OK;
}

int vo_kdtree_destroy(vo_kdtree* t) {
  if (t && ((vo_ctx_id(t->ctx) == t->ctx_id && t->ctx_id != 0) && vo_ctx_capturing(t->ctx))) return vo_fail(VO_ERR_NOT_READY, "%s: not allowed inside a graph capture", __func__);
  if (!t) return VO_OK;
  if ((vo_ctx_id(t->ctx) == t->ctx_id && t->ctx_id != 0)) {
    (void)hipSetDevice(vo_ctx_device(t->ctx));
    (void)hipStreamSynchronize(reinterpret_cast<hipStream_t>(vo_ctx_stream(t->ctx)));
  } else {
    (void)hipDeviceSynchronize();       // the context went first
  }
  for (void* p : {(void*)t->d_nodes, (void*)t->d_pts, (void*)t->d_index, t->d_q, t->d_o, t->d_i}) if (p) (void)hipFree(p);
  delete t;
  return VO_OK;
}

int vo_kdtree_info(vo_kdtree* t, int* n_points, int* n_nodes, int* n_leaves) {
  if (!t) return vo_fail(VO_ERR_INVALID_ARG, "vo_kdtree_info: null tree");
  if (n_points) *n_points = t->n;
  if (n_nodes) *n_nodes = t->n_nodes;
  if (n_leaves) *n_leaves = t->n_leaves;
  return VO_OK;
}

int vo_kdtree_best_match_fast_dev(vo_kdtree* t, const float* d_qry, int nq, float radius, int32_t* d_out) {
  if (t && !(vo_ctx_id(t->ctx) == t->ctx_id && t->ctx_id != 0)) return vo_fail(VO_ERR_INVALID_ARG, "%s: the context this tree was made on has been destroyed", __func__);
  if (!t || nq < 0 || (nq > 0 && (!d_qry || !d_out))) return vo_fail(VO_ERR_INVALID_ARG, "vo_kdtree_best_match_fast_dev: bad argument");
  if (nq == 0) return VO_OK;
  KD_CHECK(hipSetDevice(vo_ctx_device(t->ctx)));
  hipStream_t st = reinterpret_cast<hipStream_t>(vo_ctx_stream(t->ctx));
  hipLaunchKernelGGL(kd_best_fast_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, t->d_nodes, t->d_pts, t->d_index, d_qry, nq,
                     radius, d_out);
  KD_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_kdtree_best_match_fast(vo_kdtree* t, const float* qry, int nq, float radius, int32_t* out) {
  if (t && !(vo_ctx_id(t->ctx) == t->ctx_id && t->ctx_id != 0)) return vo_fail(VO_ERR_INVALID_ARG, "%s: the context this tree was made on has been destroyed", __func__);
  if (t && ((vo_ctx_id(t->ctx) == t->ctx_id && t->ctx_id != 0) && vo_ctx_capturing(t->ctx))) return vo_fail(VO_ERR_NOT_READY, "%s: not allowed inside a graph capture", __func__);
  if (!t || nq < 0 || (nq > 0 && (!qry || !out))) return vo_fail(VO_ERR_INVALID_ARG, "vo_kdtree_best_match_fast: bad argument");
  if (nq == 0) return VO_OK;
  KD_CHECK(hipSetDevice(vo_ctx_device(t->ctx)));
  hipStream_t st = reinterpret_cast<hipStream_t>(vo_ctx_stream(t->ctx));
  if (int r = kd_grow(&t->d_q, &t->q_cap, sizeof(float) * KD * (size_t)nq, st)) return r;
  if (int r = kd_grow(&t->d_o, &t->o_cap, sizeof(int32_t) * (size_t)nq, st)) return r;
  KD_CHECK(hipMemcpyAsync(t->d_q, qry, sizeof(float) * KD * (size_t)nq, hipMemcpyHostToDevice, st));
  if (int r = vo_kdtree_best_match_fast_dev(t, static_cast<const float*>(t->d_q), nq, radius, static_cast<int32_t*>(t->d_o))) return r;
  KD_CHECK(hipMemcpyAsync(out, t->d_o, sizeof(int32_t) * (size_t)nq, hipMemcpyDeviceToHost, st));
  KD_CHECK(hipStreamSynchronize(st));
  return VO_OK;
}

int vo_kdtree_fast_search_dev(vo_kdtree* t, const float* d_qry, int nq, float radius, int32_t* d_offsets, int32_t* d_indices,
                              int capacity) {
  if (t && !(vo_ctx_id(t->ctx) == t->ctx_id && t->ctx_id != 0)) return vo_fail(VO_ERR_INVALID_ARG, "%s: the context this tree was made on has been destroyed", __func__);
  if (!t || nq < 0 || capacity < 0 || !d_offsets || (nq > 0 && !d_qry) || (capacity > 0 && !d_indices))
    return vo_fail(VO_ERR_INVALID_ARG, "vo_kdtree_fast_search_dev: bad argument");
  KD_CHECK(hipSetDevice(vo_ctx_device(t->ctx)));
  hipStream_t st = reinterpret_cast<hipStream_t>(vo_ctx_stream(t->ctx));
  if (nq == 0) { KD_CHECK(hipMemsetAsync(d_offsets, 0, sizeof(int32_t), st)); return VO_OK; }
  const dim3 g((nq + 255) / 256), b(256);
  hipLaunchKernelGGL(kd_fast_search_kernel<false>, g, b, 0, st, t->d_nodes, t->d_pts, t->d_index, d_qry, nq, radius, d_offsets,
                     d_indices, capacity);
  KD_CHECK(launch_scan(st, d_offsets, nq, d_offsets + nq, nullptr, 1, 0));
  hipLaunchKernelGGL(kd_fast_search_kernel<true>, g, b, 0, st, t->d_nodes, t->d_pts, t->d_index, d_qry, nq, radius, d_offsets,
                     d_indices, capacity);
  KD_CHECK(hipGetLastError());
  return VO_OK;
}

int vo_kdtree_fast_search(vo_kdtree* t, const float* qry, int nq, float radius, int32_t* offsets, int32_t* indices, int capacity,
                          int* n_total) {
  if (t && !(vo_ctx_id(t->ctx) == t->ctx_id && t->ctx_id != 0)) return vo_fail(VO_ERR_INVALID_ARG, "%s: the context this tree was made on has been destroyed", __func__);
  if (t && ((vo_ctx_id(t->ctx) == t->ctx_id && t->ctx_id != 0) && vo_ctx_capturing(t->ctx))) return vo_fail(VO_ERR_NOT_READY, "%s: not allowed inside a graph capture", __func__);
  if (!t || nq < 0 || capacity < 0 || !offsets || !n_total || (nq > 0 && !qry) || (capacity > 0 && !indices))
    return vo_fail(VO_ERR_INVALID_ARG, "vo_kdtree_fast_search: bad argument");
  KD_CHECK(hipSetDevice(vo_ctx_device(t->ctx)));
  hipStream_t st = reinterpret_cast<hipStream_t>(vo_ctx_stream(t->ctx));
  if (int r = kd_grow(&t->d_q, &t->q_cap, sizeof(float) * KD * (size_t)(nq ? nq : 1), st)) return r;
  if (int r = kd_grow(&t->d_o, &t->o_cap, sizeof(int32_t) * ((size_t)nq + 1), st)) return r;
  if (int r = kd_grow(&t->d_i, &t->i_cap, sizeof(int32_t) * (size_t)(capacity ? capacity : 1), st)) return r;
  if (nq) KD_CHECK(hipMemcpyAsync(t->d_q, qry, sizeof(float) * KD * (size_t)nq, hipMemcpyHostToDevice, st));
  if (int r = vo_kdtree_fast_search_dev(t, static_cast<const float*>(t->d_q), nq, radius, static_cast<int32_t*>(t->d_o),
                                        static_cast<int32_t*>(t->d_i), capacity)) return r;
  KD_CHECK(hipMemcpyAsync(offsets, t->d_o, sizeof(int32_t) * ((size_t)nq + 1), hipMemcpyDeviceToHost, st));
  KD_CHECK(hipStreamSynchronize(st));
  *n_total = offsets[nq];
  if (*n_total > capacity) return vo_fail(VO_ERR_INVALID_ARG, "vo_kdtree_fast_search: %d hits, room for %d", *n_total, capacity);
  if (*n_total > 0) {
    KD_CHECK(hipMemcpyAsync(indices, t->d_i, sizeof(int32_t) * (size_t)*n_total, hipMemcpyDeviceToHost, st));
    KD_CHECK(hipStreamSynchronize(st));
  }
  return VO_OK;
}

}  // extern "C"
