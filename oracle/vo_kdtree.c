/*
 * vo_kdtree.c -- CPU restatement of the reference's actual matcher: the PCA kd-tree
 * (include/eigen_kdtree.h:6-121, split.h:8-34, eigen_covariance.h:5-43,
 * brute_force_search.h:22-41) as used by compute_correspondences_images
 * (vo_complete.cpp:12-49).  float32, single thread.
 *
 * TEST INFRASTRUCTURE ONLY (part of libvo_oracle.so): used to time the
 * reference's own algorithm as the CPU baseline of the matcher stage and to
 * check that its answers equal the exact nearest-neighbour-within-radius
 * definition the oracle and the GPU kernels implement (they must, wherever no
 * two candidates are exactly equidistant).
 *
 * Eigen's SelfAdjointEigenSolver is replaced by a cyclic Jacobi in double; the
 * split direction it returns is defined up to sign and rounding, which changes
 * which child is "left" but not the set of points visited by bestMatchFull.
 * One guard is added: the reference recurses forever when >= max_leaf points
 * fall on one side of every split (e.g. identical appearances);
 * here such a node becomes a leaf.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define DIM 11           /* slot 0: index as float (vo_complete.cpp:22), 1..10 appearance */
#define AD 10

typedef struct KdNode {
  int begin, end;                 /* range in the (reordered) point array */
  float mean[AD], normal[AD];
  struct KdNode *left, *right;
} KdNode;

static void jacobi_sym(int n, double *a, double *v) {
  for (int i = 0; i < n * n; ++i) v[i] = 0.0;
  for (int i = 0; i < n; ++i) v[i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < n; ++p) for (int q = 0; q < n; ++q) { if (p == q) diag += a[p*n+q]*a[p*n+q]; else off += a[p*n+q]*a[p*n+q]; }
    if (off <= 1e-26 * (diag + 1e-300)) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) { const double x = a[k*n+p], y = a[k*n+q]; a[k*n+p] = c*x - s*y; a[k*n+q] = s*x + c*y; }
        for (int k = 0; k < n; ++k) { const double x = a[p*n+k], y = a[q*n+k]; a[p*n+k] = c*x - s*y; a[q*n+k] = s*x + c*y; }
        for (int k = 0; k < n; ++k) { const double x = v[k*n+p], y = v[k*n+q]; v[k*n+p] = c*x - s*y; v[k*n+q] = s*x + c*y; }
      }
  }
}

/* computeMeanAndCovariance (eigen_covariance.h:5-30) + largestEigenVector (:35-43) */
static void mean_and_direction(const float *pts, int begin, int end, float *mean, float *normal) {
  float m[AD], cov[AD * AD];
  for (int i = 0; i < AD; ++i) m[i] = 0.f;
  for (int i = 0; i < AD * AD; ++i) cov[i] = 0.f;
  int k = 0;
  for (int it = begin; it < end; ++it) {
    const float *v = pts + (size_t)it * DIM + 1;
    for (int i = 0; i < AD; ++i) m[i] += v[i];
    for (int i = 0; i < AD; ++i) for (int j = 0; j < AD; ++j) cov[i * AD + j] += v[i] * v[j];
    ++k;
  }
  const float ik = (float)(1.0 / k);
  for (int i = 0; i < AD; ++i) m[i] *= ik;
  for (int i = 0; i < AD * AD; ++i) cov[i] *= ik;
  for (int i = 0; i < AD; ++i) for (int j = 0; j < AD; ++j) cov[i * AD + j] -= m[i] * m[j];
  const float sc = (float)k / (float)(k - 1);
  double a[AD * AD], v[AD * AD];
  for (int i = 0; i < AD * AD; ++i) a[i] = (double)(cov[i] * sc);
  jacobi_sym(AD, a, v);
  int best = 0;
  for (int i = 1; i < AD; ++i) if (a[i * AD + i] > a[best * AD + best]) best = i;
  /* sign convention (shared with csrc/kdtree.hip): the component of largest magnitude is positive, the first among
   * equals; Eigen's solver has no such rule -- the sign decides which child is "left", i.e. the order inside a leaf */
  int big = 0;
  for (int i = 1; i < AD; ++i) if (fabs(v[i * AD + best]) > fabs(v[big * AD + best])) big = i;
  const double sgn = v[big * AD + best] < 0.0 ? -1.0 : 1.0;
  for (int i = 0; i < AD; ++i) { mean[i] = m[i]; normal[i] = (float)(sgn * v[i * AD + best]); }
}

static float plane_dist(const float *p11, const float *mean, const float *normal) {
  float s = 0.f;
  for (int i = 0; i < AD; ++i) s += (p11[1 + i] - mean[i]) * normal[i];
  return s;
}

/* split (split.h:8-34): predicate-true items first; returns the middle index */
static int split_range(float *pts, int begin, int end, const float *mean, const float *normal) {
  int lower = begin, upper = end;           /* upper.base() */
  float tmp[DIM];
  while (lower != upper) {
    float *vl = pts + (size_t)lower * DIM;
    if (plane_dist(vl, mean, normal) < 0.f) {
      ++lower;
    } else {
      float *vu = pts + (size_t)(upper - 1) * DIM;
      memcpy(tmp, vl, sizeof(tmp)); memcpy(vl, vu, sizeof(tmp)); memcpy(vu, tmp, sizeof(tmp));
      --upper;
    }
  }
  return upper;
}

static KdNode *build(float *pts, int begin, int end, int max_leaf) {       /* eigen_kdtree.h:18-38 */
  KdNode *nd = (KdNode *)calloc(1, sizeof(KdNode));
  nd->begin = begin; nd->end = end;
  if (end - begin < max_leaf) return nd;
  mean_and_direction(pts, begin, end, nd->mean, nd->normal);
  const int middle = split_range(pts, begin, end, nd->mean, nd->normal);
  if (middle == begin || middle == end) return nd;                         /* guard, see header */
  nd->left = build(pts, begin, middle, max_leaf);
  nd->right = build(pts, middle, end, max_leaf);
  return nd;
}

static void free_tree(KdNode *n) { if (!n) return; free_tree(n->left); free_tree(n->right); free(n); }

static float sqdist(const float *p, const float *q) {
  float s = 0.f;
  for (int i = 1; i < DIM; ++i) { const float d = p[i] - q[i]; s += d * d; }
  return s;
}

/* bruteForceBestMatch (brute_force_search.h:22-41) */
static float *leaf_best(float *pts, int begin, int end, const float *query, float norm) {
  float *best = 0;
  float best_sq = norm * norm;
  for (int it = begin; it < end; ++it) {
    float *p = pts + (size_t)it * DIM;
    const float d = sqdist(p, query);
    if (d < best_sq) { best = p; best_sq = d; }
  }
  return best;
}

/* bestMatchFull (eigen_kdtree.h:90-115) */
static float *best_match_full(KdNode *n, float *pts, const float *query, float norm) {
  if (!n->left && !n->right) return leaf_best(pts, n->begin, n->end, query, norm);
  const float d = plane_dist(query, n->mean, n->normal);
  if (d < -norm) return best_match_full(n->left, pts, query, norm);
  if (d > norm) return best_match_full(n->right, pts, query, norm);
  float *pl = best_match_full(n->left, pts, query, norm);
  float *pr = best_match_full(n->right, pts, query, norm);
  float dl = norm * norm, dr = norm * norm;
  if (pl) dl = sqdist(pl, query);
  if (pr) dr = sqdist(pr, query);
  if (dl < dr) return pl;
  return pr;
}

/* compute_correspondences_images with the kd-tree (vo_complete.cpp:12-49).
 * build_seconds / query_seconds (may be null) receive the two phases' wall time. */
#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int vo32_match_kdtree(const float *a1, int n1, const float *a2, int n2, float radius, int max_leaf, int *out_pairs,
                      double *build_seconds, double *query_seconds) {
  const int tree_is_1 = (n1 >= n2);
  const float *ta = tree_is_1 ? a1 : a2, *qa = tree_is_1 ? a2 : a1;
  const int nt = tree_is_1 ? n1 : n2, nq = tree_is_1 ? n2 : n1;
  float *pts = (float *)malloc(sizeof(float) * DIM * (size_t)(nt > 0 ? nt : 1));
  for (int i = 0; i < nt; ++i) { pts[(size_t)i * DIM] = (float)i; memcpy(pts + (size_t)i * DIM + 1, ta + (size_t)i * AD, sizeof(float) * AD); }
  double t0 = now_s();
  KdNode *root = build(pts, 0, nt, max_leaf);
  double t1 = now_s();
  int n_out = 0;
  float q[DIM];
  for (int i = 0; i < nq; ++i) {
    q[0] = (float)i;
    memcpy(q + 1, qa + (size_t)i * AD, sizeof(float) * AD);
    float *m = best_match_full(root, pts, q, radius);
    if (m) {
      if (tree_is_1) { out_pairs[2 * n_out] = (int)m[0]; out_pairs[2 * n_out + 1] = i; }
      else { out_pairs[2 * n_out] = i; out_pairs[2 * n_out + 1] = (int)m[0]; }
      n_out++;
    }
  }
  double t2 = now_s();
  if (build_seconds) *build_seconds = t1 - t0;
  if (query_seconds) *query_seconds = t2 - t1;
  free_tree(root);
  free(pts);
  return n_out;
}

/* ---- TreeNode_::fullSearch, eigen_kdtree.h:56-71 (+ bruteForceSearch, brute_force_search.h:3-20) ---- */
/* Appends to out[] (room for cap ints) the stored index of every point within `norm`, in the reference's
 * traversal order; returns the number of matches (it keeps counting past cap). */
static int full_search(const KdNode *n, const float *pts, const float *query, float norm, int *out, int cap, int have) {
  if (!n->left && !n->right) {
    const float sq = norm * norm;                                       /* brute_force_search.h:10 */
    for (int i = n->begin; i < n->end; ++i) {
      if (sqdist(pts + (size_t)i * DIM, query) < sq) {                  /* :14 */
        if (have < cap) out[have] = (int)pts[(size_t)i * DIM];
        ++have;
      }
    }
    return have;
  }
  const float d = plane_dist(query, n->mean, n->normal);
  if (d < -norm) return full_search(n->left, pts, query, norm, out, cap, have);            /* :64-65 */
  if (d > norm) return full_search(n->right, pts, query, norm, out, cap, have);            /* :66-67 */
  have = full_search(n->left, pts, query, norm, out, cap, have);                           /* :69 */
  return full_search(n->right, pts, query, norm, out, cap, have);                          /* :70 */
}

/* CSR radius search of every query against the tree set.  brute != 0: plain double loop (the arbiter for the
 * kd-tree restatement).  offsets[nq+1]; indices[cap]; returns the total number of matches. */
int vo32_radius_search(const float *tree_app, int nt, const float *query_app, int nq, float radius, int max_leaf,
                       int brute, int *offsets, int *indices, int cap) {
  float *pts = (float *)malloc(sizeof(float) * DIM * (size_t)(nt > 0 ? nt : 1));
  for (int i = 0; i < nt; ++i) { pts[(size_t)i * DIM] = (float)i; memcpy(pts + (size_t)i * DIM + 1, tree_app + (size_t)i * AD, sizeof(float) * AD); }
  KdNode *root = (!brute && nt > 0) ? build(pts, 0, nt, max_leaf) : 0;
  int total = 0;
  float q[DIM];
  for (int i = 0; i < nq; ++i) {
    offsets[i] = total;
    q[0] = (float)i;
    memcpy(q + 1, query_app + (size_t)i * AD, sizeof(float) * AD);
    if (root) {
      total = full_search(root, pts, q, radius, indices, cap, total);
    } else {
      const float sq = radius * radius;
      for (int j = 0; j < nt; ++j)
        if (sqdist(pts + (size_t)j * DIM, q) < sq) { if (total < cap) indices[total] = j; ++total; }
    }
  }
  offsets[nq] = total;
  if (root) free_tree(root);
  free(pts);
  return total;
}

/* ---- the approximate modes: bestMatchFast (eigen_kdtree.h:75-85) and fastSearch (:40-52) ---- */
/* Both descend one side of every split plane (distance < 0 -> left) and brute-force the leaf.  Their answers depend on
 * the tree: on the split directions (here: the Jacobi stand-in for Eigen's eigen-solver) and on the order split()
 * leaves the points in.  out_index[i] = stored index of the best leaf point within the radius, or -1. */
static const KdNode *leaf_of(const KdNode *n, const float *query) {
  while (n->left || n->right) n = plane_dist(query, n->mean, n->normal) < 0.f ? n->left : n->right;   /* :47-51, :80-84 */
  return n;
}

int vo32_kdtree_fast(const float *tree_app, int nt, const float *query_app, int nq, float radius, int max_leaf,
                     int *out_best, int *offsets, int *indices, int cap, int *n_nodes_out) {
  float *pts = (float *)malloc(sizeof(float) * DIM * (size_t)(nt > 0 ? nt : 1));
  for (int i = 0; i < nt; ++i) { pts[(size_t)i * DIM] = (float)i; memcpy(pts + (size_t)i * DIM + 1, tree_app + (size_t)i * AD, sizeof(float) * AD); }
  KdNode *root = build(pts, 0, nt, max_leaf);
  int total = 0;
  float q[DIM];
  for (int i = 0; i < nq; ++i) {
    q[0] = (float)i;
    memcpy(q + 1, query_app + (size_t)i * AD, sizeof(float) * AD);
    const KdNode *lf = leaf_of(root, q);
    if (out_best) {
      float *m = leaf_best(pts, lf->begin, lf->end, q, radius);                  /* bestMatchFast */
      out_best[i] = m ? (int)m[0] : -1;
    }
    if (offsets) {                                                               /* fastSearch */
      offsets[i] = total;
      const float sq = radius * radius;
      for (int j = lf->begin; j < lf->end; ++j)
        if (sqdist(pts + (size_t)j * DIM, q) < sq) { if (total < cap) indices[total] = (int)pts[(size_t)j * DIM]; ++total; }
    }
  }
  if (offsets) offsets[nq] = total;
  if (n_nodes_out) {                      /* node count, for a structural comparison with the product's tree */
    int count = 0;
    const KdNode *stack[256]; int sp = 0;
    stack[sp++] = root;
    while (sp) { const KdNode *n = stack[--sp]; ++count; if (n->left) stack[sp++] = n->left; if (n->right) stack[sp++] = n->right; }
    *n_nodes_out = count;
  }
  free_tree(root);
  free(pts);
  return total;
}
