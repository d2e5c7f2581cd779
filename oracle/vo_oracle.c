/*
 * vo_oracle.c -- CPU oracle for the projective-ICP hot path of
 * lucanunz/Visual-odometry (picp_solver + appearance matcher + triangulation
 * + the glue between them).  Plain C, scalar, single thread.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product (libvo_hip.so) never links,
 * loads or calls anything in this directory.
 *
 * PARITY PINNED BY KNOWN ANSWERS, UNPINNED AT THE BIT LEVEL: see vo_oracle_impl.h.
 * The reference's data directory holds the ground truth and its README the
 * end-to-end metrics (README.md:74-79); oracle/vo_pipeline.py reproduces both.
 *
 * Build: make -C oracle   (gcc -O3 -ffp-contract=off, no -march: mirrors the
 * reference's "-O3 -DNDEBUG" x86-64 baseline build, CMakeLists.txt:6-7)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define PREFIX vo32_
#include "vo_oracle_impl.h"
#undef REAL
#undef PREFIX

#define REAL double
#define PREFIX vo64_
#include "vo_oracle_impl.h"
#undef REAL
#undef PREFIX

/* ---- extract_correspondences_world, vo_complete.cpp:52-66 -------------- */
/* Literal O(C_img * C_world) form: for each image pair in order, the FIRST
 * world pair whose .first equals the image pair's .first. */
int vo_join(const int *img, int n_img, const int *world, int n_world, int *out) {
  int n_out = 0;
  for (int i = 0; i < n_img; ++i) {
    const int idx_ref = img[2 * i];                               /* :56 */
    for (int j = 0; j < n_world; ++j) {
      if (world[2 * j] == idx_ref) {                              /* :58 */
        out[2 * n_out] = img[2 * i + 1];                          /* :59 */
        out[2 * n_out + 1] = world[2 * j + 1];
        n_out++;
        break;
      }
    }
  }
  return n_out;
}

/* Same result in O(C): first-occurrence table over ref indices.  Used as the
 * "sane" CPU baseline; checked against vo_join in the tests. */
int vo_join_linear(const int *img, int n_img, const int *world, int n_world, int *out) {
  int max_ref = -1;
  for (int j = 0; j < n_world; ++j) if (world[2 * j] > max_ref) max_ref = world[2 * j];
  int *first = (int *)malloc(sizeof(int) * (size_t)(max_ref + 2));
  for (int i = 0; i <= max_ref; ++i) first[i] = -1;
  for (int j = n_world - 1; j >= 0; --j) if (world[2 * j] >= 0) first[world[2 * j]] = j;
  int n_out = 0;
  for (int i = 0; i < n_img; ++i) {
    const int r = img[2 * i];
    if (r < 0 || r > max_ref) continue;
    const int j = first[r];
    if (j < 0) continue;
    out[2 * n_out] = img[2 * i + 1];
    out[2 * n_out + 1] = world[2 * j + 1];
    n_out++;
  }
  free(first);
  return n_out;
}

/* ---- all-cores CPU baseline (SURVEY 8(d) "CPU baseline timing" (ii)) ------------------------ */
/* The reference is single-threaded; this is the "strong" baseline only: the same linearize loop
 * (vo32_picp_linearize) over n_threads contiguous chunks with per-thread H/b/chi partials, summed
 * in thread order (deterministic for a fixed n_threads; n_threads = 1 is bit-identical to
 * vo32_picp_one_round), followed by the serial tail of oneRound (picp_solver.cpp:102-110).
 * Returns the number of threads used. */
#ifdef _OPENMP
#include <omp.h>
#endif
int vo32_picp_solve_mt(vo32_picp *s, const int *corr, int n, int keep_outliers, int n_iters, int n_threads) {
  if (n_threads < 1) n_threads = 1;
#ifndef _OPENMP
  n_threads = 1;
#endif
  vo32_picp *part = (vo32_picp *)malloc(sizeof(vo32_picp) * (size_t)n_threads);
  if (!part) return 0;
  for (int it = 0; it < n_iters; ++it) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
#endif
    for (int t = 0; t < n_threads; ++t) {
      vo32_picp mine = *s;                       /* on the thread's own stack: no false sharing */
      const int lo = (int)((long long)n * t / n_threads), hi = (int)((long long)n * (t + 1) / n_threads);
      vo32_picp_linearize(&mine, corr + 2 * lo, hi - lo, keep_outliers);
      part[t] = mine;
    }
    for (int i = 0; i < 36; ++i) s->H[i] = 0.f;
    for (int i = 0; i < 6; ++i) s->b[i] = 0.f;
    s->chi_inliers = s->chi_outliers = 0.f;
    s->num_inliers = 0;
    for (int t = 0; t < n_threads; ++t) {
      for (int i = 0; i < 36; ++i) s->H[i] += part[t].H[i];
      for (int i = 0; i < 6; ++i) s->b[i] += part[t].b[i];
      s->chi_inliers += part[t].chi_inliers;
      s->chi_outliers += part[t].chi_outliers;
      s->num_inliers += part[t].num_inliers;
    }
    for (int i = 0; i < 6; ++i) s->H[i + 6 * i] += 1.f * s->damping;
    if (s->num_inliers < s->min_num_inliers) continue;
    float nb[6], dx[6], dT[16];
    for (int i = 0; i < 6; ++i) nb[i] = -s->b[i];
    vo32_ldlt_solve(6, s->H, nb, dx);
    vo32_v2t_euler(dx, dT);
    vo32_iso_mul(dT, s->cam.T, s->cam.T);
  }
  free(part);
  return n_threads;
}

int vo_oracle_abi_version(void) { return 2; }
