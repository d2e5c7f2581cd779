/*
 * vo_oracle.c -- CPU oracle for the projective-ICP hot path of
 * lucanunz/Visual-odometry (picp_solver + appearance matcher + triangulation
 * + the glue between them).  Plain C, scalar, single thread.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product (libvo_hip.so) never links,
 * loads or calls anything in this directory.
 *
 * PARITY UNPINNED (kernel level): see vo_oracle_impl.h.  The only numbers the
 * reference publishes for this path are the end-to-end README metrics on
 * example_data (README.md:74-79); oracle/vo_pipeline.py reproduces them.
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

int vo_oracle_abi_version(void) { return 1; }
