/*
 * vo_oracle_impl.h -- body of the CPU oracle, included twice by vo_oracle.c:
 * once with REAL=float (prefix vo32_, "ref32": the reference's own arithmetic
 * type) and once with REAL=double (prefix vo64_, "ref64": the arbiter).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is a product path; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY PINNED BY KNOWN ANSWERS, UNPINNED AT THE BIT LEVEL.  The reference
 * (lucanunz/Visual-odometry) ships no golden vectors / assertions and cannot be
 * compiled here (needs Eigen3, un-vendored, version unpinned, absent from the
 * image), so no output of a reference BUILD exists to compare bits with.  What
 * the reference does hold is its data directory with the ground truth
 * (trajectory.dat, world.dat, landmark ids per measurement) and the test
 * programs that run this path on it; the oracle is checked against those
 * known answers (tests/test_known_answers_cpu.py, oracle/vo_pipeline.py):
 *   picp_real_data_allKnown.cpp  landmarks + association known: all 121 camera
 *                                poses equal the ground truth to < 1e-4 (4.7e-5)
 *   initialization_real_data.cpp first relative pose: rotation 3e-6, translation
 *                                direction 6e-6, triangulated landmarks = world.dat
 *   vo_daKnown.cpp / vo_complete scale 1/r_t = 0.47336 vs README 0.47337, and
 *                                the README metrics (tests/test_vo_complete_cpu.py)
 *   matcher                      pairs = the landmark-id overlap of the frames
 * This file restates the reference's algorithm, sequential float arithmetic,
 * one correspondence after the other, each function citing the file:line it
 * follows under /root/reference.  Where the arithmetic lives inside Eigen
 * (small fixed-size products, ldlt, inverse) the published Eigen 3.4 algorithm
 * is restated and the association order chosen is written next to the code.
 *
 * Conventions: matrices are column-major like Eigen's defaults
 * (K[r+3c], T[r+4c]); index pairs are int32 (first, second) packed.
 * Compile with -ffp-contract=off (the reference builds with plain -O3, SSE2
 * baseline, i.e. no FMA contraction: CMakeLists.txt:6-7).
 */

#ifndef REAL
#error "include from vo_oracle.c"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(PREFIX, name)

typedef struct {
  int rows, cols, z_near, z_far; /* ints, as camera.h:56-59 */
  REAL K[9];                     /* camera matrix, col-major */
  REAL T[16];                    /* world_in_camera_pose, col-major 4x4 */
} FN(camera);

typedef struct {
  FN(camera) cam;                /* picp_solver.h:64 (state) */
  REAL kernel_threshold;         /* :65, default 1000 (picp_solver.cpp:13) */
  REAL damping;                  /* :66, 1 (picp_solver.cpp:10) */
  int min_num_inliers;           /* :67, 0 (picp_solver.cpp:11) */
  const REAL *world;             /* :68 borrowed, xyz packed */
  const REAL *meas;              /* :69 borrowed, uv packed */
  REAL H[36];                    /* :70 col-major 6x6 */
  REAL b[6];                     /* :71 */
  REAL chi_inliers, chi_outliers;/* :72-73 */
  int num_inliers;               /* :74 */
} FN(picp);

/* Eigen 3.4 evaluates a fixed-size, non-vectorisable inner product through
 * redux_novec_unroller (Core/Redux.h): a length-3 sum is x0 + (x1 + x2),
 * a length-2 sum x0 + x1.  Every small product below uses these helpers. */
static inline REAL FN(dot3)(REAL a0, REAL b0, REAL a1, REAL b1, REAL a2, REAL b2) {
  return a0 * b0 + (a1 * b1 + a2 * b2);
}

/* y = M(3x3, col-major, leading dim ld) * x */
static inline void FN(mat3_vec)(const REAL *M, int ld, const REAL x[3], REAL y[3]) {
  for (int i = 0; i < 3; ++i)
    y[i] = FN(dot3)(M[i], x[0], M[i + ld], x[1], M[i + 2 * ld], x[2]);
}

/* Isometry3f * Vector3f: Eigen's transform_right_product_impl builds
 * res = translation, then res += linear * v (Geometry/Transform.h). */
static inline void FN(iso_point)(const REAL T[16], const REAL p[3], REAL out[3]) {
  REAL rp[3];
  FN(mat3_vec)(T, 4, p, rp);
  for (int i = 0; i < 3; ++i) out[i] = T[12 + i] + rp[i];
}

/* C(3x3) = A(3x3) * B(3x3), all col-major with leading dims la/lb/lc */
static inline void FN(mat3_mul)(const REAL *A, int la, const REAL *B, int lb, REAL *C, int lc) {
  REAL tmp[9];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      tmp[r + 3 * c] = FN(dot3)(A[r], B[lb * c], A[r + la], B[1 + lb * c], A[r + 2 * la], B[2 + lb * c]);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) C[r + lc * c] = tmp[r + 3 * c];
}

/* ---- Camera::projectPoint, camera.h:25-37 ------------------------------ */
int FN(project_point)(const FN(camera) * cam, const REAL p[3], REAL uv[2]) {
  REAL pc[3];
  FN(iso_point)(cam->T, p, pc);                                   /* :27 */
  if (pc[2] > (REAL)cam->z_far || pc[2] < (REAL)cam->z_near)      /* :28 */
    return 0;
  REAL ph[3];
  FN(mat3_vec)(cam->K, 3, pc, ph);                                /* :30 */
  /* :31  head<2>() * (1./z): the reciprocal is taken in double and rounded
   * to the scalar type before the vector multiply. */
  const REAL inv = (REAL)(1.0 / (double)ph[2]);
  uv[0] = ph[0] * inv;
  uv[1] = ph[1] * inv;
  if (uv[0] < (REAL)0 || uv[0] > (REAL)(cam->cols - 1)) return 0; /* :32 */
  if (uv[1] < (REAL)0 || uv[1] > (REAL)(cam->rows - 1)) return 0; /* :34 */
  return 1;
}

/* ---- Camera::projectPoints, camera.cpp:16-37 --------------------------- */
/* out_uv has room for n points; *n_out = final size; returns #inside. */
int FN(project_points)(const FN(camera) * cam, const REAL *world, int n, int keep_indices,
                       REAL *out_uv, int *n_out) {
  int num_image_points = 0, num_points_inside = 0;
  for (int i = 0; i < n; ++i) {
    REAL *uv = out_uv + 2 * num_image_points;                     /* :25 */
    int inside = FN(project_point)(cam, world + 3 * i, uv);
    if (inside) num_points_inside++;
    else { uv[0] = (REAL)-1; uv[1] = (REAL)-1; }                  /* :30 */
    if (keep_indices || inside) num_image_points++;               /* :31 */
  }
  *n_out = num_image_points;
  return num_points_inside;
}

/* ---- PICPSolver::errorAndJacobian, picp_solver.cpp:25-53 --------------- */
/* J is 2x6 col-major (J[r+2c]). */
int FN(error_and_jacobian)(const FN(camera) * cam, const REAL wp[3], const REAL z[2], REAL e[2],
                           REAL J[12]) {
  REAL uv[2];
  if (!FN(project_point)(cam, wp, uv)) return 0;                  /* :32-34 */
  e[0] = uv[0] - z[0];                                            /* :35 */
  e[1] = uv[1] - z[1];
  REAL pc[3];
  FN(iso_point)(cam->T, wp, pc);                                  /* :38 */
  /* Jr = [I3 | skew(-pc)], :39-41 with skew from utils.h:96-102 */
  REAL Jr[18];
  for (int i = 0; i < 18; ++i) Jr[i] = (REAL)0;
  Jr[0 + 3 * 0] = (REAL)1; Jr[1 + 3 * 1] = (REAL)1; Jr[2 + 3 * 2] = (REAL)1;
  const REAL v0 = -pc[0], v1 = -pc[1], v2 = -pc[2];
  /* skew(v) = [0 -v2 v1; v2 0 -v0; -v1 v0 0] into columns 3..5 */
  Jr[0 + 3 * 3] = (REAL)0; Jr[0 + 3 * 4] = -v2;      Jr[0 + 3 * 5] = v1;
  Jr[1 + 3 * 3] = v2;      Jr[1 + 3 * 4] = (REAL)0; Jr[1 + 3 * 5] = -v0;
  Jr[2 + 3 * 3] = -v1;     Jr[2 + 3 * 4] = v0;      Jr[2 + 3 * 5] = (REAL)0;
  REAL ph[3];
  FN(mat3_vec)(cam->K, 3, pc, ph);                                /* :43 */
  const REAL iz = (REAL)(1.0 / (double)ph[2]);                    /* :44 */
  const REAL iz2 = iz * iz;                                       /* :45 */
  REAL Jp[6];                                                     /* :47-49, 2x3 col-major */
  Jp[0 + 2 * 0] = iz;      Jp[0 + 2 * 1] = (REAL)0; Jp[0 + 2 * 2] = -ph[0] * iz2;
  Jp[1 + 2 * 0] = (REAL)0; Jp[1 + 2 * 1] = iz;      Jp[1 + 2 * 2] = -ph[1] * iz2;
  /* :51  (Jp*K) is evaluated into a 2x3 temporary, then times Jr */
  REAL JpK[6];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 2; ++r)
      JpK[r + 2 * c] = FN(dot3)(Jp[r], cam->K[3 * c], Jp[r + 2], cam->K[1 + 3 * c], Jp[r + 4], cam->K[2 + 3 * c]);
  for (int c = 0; c < 6; ++c)
    for (int r = 0; r < 2; ++r)
      J[r + 2 * c] = FN(dot3)(JpK[r], Jr[3 * c], JpK[r + 2], Jr[1 + 3 * c], JpK[r + 4], Jr[2 + 3 * c]);
  return 1;
}

/* ---- PICPSolver::linearize, picp_solver.cpp:55-96 ---------------------- */
void FN(picp_linearize)(FN(picp) * s, const int *corr, int n, int keep_outliers) {
  for (int i = 0; i < 36; ++i) s->H[i] = (REAL)0;                 /* :57-61 */
  for (int i = 0; i < 6; ++i) s->b[i] = (REAL)0;
  s->num_inliers = 0;
  s->chi_inliers = (REAL)0;
  s->chi_outliers = (REAL)0;
  for (int k = 0; k < n; ++k) {                                   /* :62 */
    const int ref_idx = corr[2 * k];                              /* :66 .first -> measurement */
    const int curr_idx = corr[2 * k + 1];                         /* :67 .second -> world */
    REAL e[2], J[12];
    if (!FN(error_and_jacobian)(&s->cam, s->world + 3 * curr_idx, s->meas + 2 * ref_idx, e, J))
      continue;                                                   /* :72-73 */
    const REAL chi = e[0] * e[0] + e[1] * e[1];                   /* :75 */
    REAL lambda = (REAL)1;
    int is_inlier = 1;
    if (chi > s->kernel_threshold) {                              /* :78 */
      lambda = (REAL)sqrt((double)(s->kernel_threshold / chi));   /* :80 */
      is_inlier = 0;
      s->chi_outliers += chi;
    } else {
      s->chi_inliers += chi;                                      /* :86-87 */
      s->num_inliers++;
    }
    if (is_inlier || keep_outliers) {                             /* :90-94 */
      for (int c = 0; c < 6; ++c)
        for (int r = 0; r < 6; ++r) {
          const REAL jtj = J[0 + 2 * r] * J[0 + 2 * c] + J[1 + 2 * r] * J[1 + 2 * c];
          s->H[r + 6 * c] += jtj * lambda;
        }
      for (int r = 0; r < 6; ++r) {
        const REAL jte = J[0 + 2 * r] * e[0] + J[1 + 2 * r] * e[1];
        s->b[r] += jte * lambda;
      }
    }
  }
}

/* ---- Eigen::LDLT<Matrix,Lower> (Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked
 * + LDLT::_solve_impl), restated for a dense col-major n x n, n <= 6.
 * Used at picp_solver.cpp:109 (n=6) and utils.cpp:40 (n=2).
 * Pivot = largest |diagonal| of the trailing block, first one on ties;
 * rank-k updates are applied one column after the other, left to right. */
void FN(ldlt_solve)(int n, const REAL *A_in, const REAL *rhs, REAL *x) {
  REAL m[36], tmp[6];
  int tr[6];
  for (int i = 0; i < n * n; ++i) m[i] = A_in[i];
#define M(r, c) m[(r) + n * (c)]
  int all_zero = 0;
  for (int k = 0; k < n && !all_zero; ++k) {
    int piv = k;
    REAL big = (REAL)fabs((double)M(k, k));
    for (int i = k + 1; i < n; ++i) {
      const REAL a = (REAL)fabs((double)M(i, i));
      if (a > big) { big = a; piv = i; }
    }
    tr[k] = piv;
    if (piv != k) {
      const int s = n - piv - 1;
      for (int j = 0; j < k; ++j) { REAL t = M(k, j); M(k, j) = M(piv, j); M(piv, j) = t; }
      for (int j = 0; j < s; ++j) {
        REAL t = M(piv + 1 + j, k); M(piv + 1 + j, k) = M(piv + 1 + j, piv); M(piv + 1 + j, piv) = t;
      }
      { REAL t = M(k, k); M(k, k) = M(piv, piv); M(piv, piv) = t; }
      for (int i = k + 1; i < piv; ++i) { REAL t = M(i, k); M(i, k) = M(piv, i); M(piv, i) = t; }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; ++j) tmp[j] = M(j, j) * M(k, j);
      REAL acc = (REAL)0;
      for (int j = 0; j < k; ++j) acc += M(k, j) * tmp[j];
      M(k, k) -= acc;
      for (int i = 0; i < rs; ++i) {
        REAL a = (REAL)0;
        for (int j = 0; j < k; ++j) a += M(k + 1 + i, j) * tmp[j];
        M(k + 1 + i, k) -= a;
      }
    }
    const REAL akk = M(k, k);
    const int pivot_ok = fabs((double)akk) > 0.0;
    if (k == 0 && !pivot_ok) {
      /* whole matrix is zero: Eigen fills the transpositions and stops */
      for (int j = 0; j < n; ++j) tr[j] = j;
      all_zero = 1;
      break;
    }
    if (rs > 0 && pivot_ok)
      for (int i = 0; i < rs; ++i) M(k + 1 + i, k) /= akk;
  }
  /* solve: x = P^T L^-T D^+ L^-1 P rhs */
  REAL y[6];
  for (int i = 0; i < n; ++i) y[i] = rhs[i];
  for (int k = 0; k < n; ++k)
    if (tr[k] != k) { REAL t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < n; ++i) {        /* unit-lower forward substitution */
    REAL acc = y[i];
    for (int j = 0; j < i; ++j) acc -= M(i, j) * y[j];
    y[i] = acc;
  }
  /* pseudo-inverse of D with Eigen's tolerance = smallest normalised value */
  const REAL tol = (sizeof(REAL) == 4) ? (REAL)1.17549435e-38 : (REAL)2.2250738585072014e-308;
  for (int i = 0; i < n; ++i) {
    if (fabs((double)M(i, i)) > (double)tol) y[i] /= M(i, i);
    else y[i] = (REAL)0;
  }
  for (int i = n - 1; i >= 0; --i) {   /* unit-upper (L^T) back substitution */
    REAL acc = y[i];
    for (int j = i + 1; j < n; ++j) acc -= M(j, i) * y[j];
    y[i] = acc;
  }
  for (int k = n - 1; k >= 0; --k)
    if (tr[k] != k) { REAL t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < n; ++i) x[i] = y[i];
#undef M
}

/* ---- v2tEuler / Rotation{X,Y,Z}, utils.h:16-78 ------------------------- */
/* sin/cos bind to the double libm overloads and are rounded to the scalar. */
void FN(v2t_euler)(const REAL v[6], REAL T[16]) {
  const REAL sx = (REAL)sin((double)v[3]), cx = (REAL)cos((double)v[3]);
  const REAL sy = (REAL)sin((double)v[4]), cy = (REAL)cos((double)v[4]);
  const REAL sz = (REAL)sin((double)v[5]), cz = (REAL)cos((double)v[5]);
  const REAL one = (REAL)1, zero = (REAL)0;
  /* col-major 3x3 */
  const REAL Rx[9] = {one, zero, zero, zero, cx, sx, zero, -sx, cx};  /* :22-25 */
  const REAL Ry[9] = {cy, zero, -sy, zero, one, zero, sy, zero, cy};  /* :38-41 */
  const REAL Rz[9] = {cz, sz, zero, -sz, cz, zero, zero, zero, one};  /* :54-57 */
  REAL Rxy[9], R[9];
  FN(mat3_mul)(Rx, 3, Ry, 3, Rxy, 3);                                 /* :65 */
  FN(mat3_mul)(Rxy, 3, Rz, 3, R, 3);
  for (int i = 0; i < 16; ++i) T[i] = zero;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) T[r + 4 * c] = R[r + 3 * c];
  T[12] = v[0]; T[13] = v[1]; T[14] = v[2]; T[15] = one;              /* :76 */
}

/* Isometry * Isometry: linear = La*Lb ; translation = La*tb + ta */
void FN(iso_mul)(const REAL A[16], const REAL B[16], REAL C[16]) {
  REAL out[16];
  for (int i = 0; i < 16; ++i) out[i] = (REAL)0;
  FN(mat3_mul)(A, 4, B, 4, out, 4);
  REAL rt[3];
  FN(mat3_vec)(A, 4, B + 12, rt);
  for (int i = 0; i < 3; ++i) out[12 + i] = rt[i] + A[12 + i];
  out[15] = (REAL)1;
  for (int i = 0; i < 16; ++i) C[i] = out[i];
}

/* ---- PICPSolver::oneRound, picp_solver.cpp:98-112 ---------------------- */
int FN(picp_one_round)(FN(picp) * s, const int *corr, int n, int keep_outliers) {
  FN(picp_linearize)(s, corr, n, keep_outliers);                  /* :101 */
  for (int i = 0; i < 6; ++i) s->H[i + 6 * i] += (REAL)1 * s->damping; /* :102 */
  if (s->num_inliers < s->min_num_inliers) return 0;              /* :103-107 */
  REAL nb[6], dx[6], dT[16];
  for (int i = 0; i < 6; ++i) nb[i] = -s->b[i];
  FN(ldlt_solve)(6, s->H, nb, dx);                                /* :109 */
  FN(v2t_euler)(dx, dT);
  FN(iso_mul)(dT, s->cam.T, s->cam.T);                            /* :110 */
  return 1;
}

void FN(picp_init)(FN(picp) * s, const FN(camera) * cam, const REAL *world, const REAL *meas) {
  s->cam = *cam;                                                  /* picp_solver.cpp:20-22 */
  s->world = world;
  s->meas = meas;
}

void FN(picp_ctor)(FN(picp) * s) {                                /* picp_solver.cpp:6-14 */
  s->world = 0; s->meas = 0;
  s->damping = (REAL)1;
  s->min_num_inliers = 0;
  s->num_inliers = 0;
  s->kernel_threshold = (REAL)1000;
  s->chi_inliers = s->chi_outliers = (REAL)0;
  for (int i = 0; i < 36; ++i) s->H[i] = (REAL)0;
  for (int i = 0; i < 6; ++i) s->b[i] = (REAL)0;
}

/* Convenience for tests: run n_iters rounds, recording H (before damping),
 * b, stats and the pose after each round if the trace pointers are non-null. */
void FN(picp_solve)(FN(picp) * s, const int *corr, int n, int keep_outliers, int n_iters,
                    REAL *trace_H, REAL *trace_b, REAL *trace_stats, REAL *trace_T) {
  for (int it = 0; it < n_iters; ++it) {
    FN(picp_one_round)(s, corr, n, keep_outliers);
    if (trace_H) {
      for (int i = 0; i < 36; ++i) trace_H[36 * it + i] = s->H[i];
      for (int i = 0; i < 6; ++i) trace_H[36 * it + 7 * i] -= (REAL)1 * s->damping;
    }
    if (trace_b) for (int i = 0; i < 6; ++i) trace_b[6 * it + i] = s->b[i];
    if (trace_stats) {
      trace_stats[3 * it + 0] = s->chi_inliers;
      trace_stats[3 * it + 1] = s->chi_outliers;
      trace_stats[3 * it + 2] = (REAL)s->num_inliers;
    }
    if (trace_T) for (int i = 0; i < 16; ++i) trace_T[16 * it + i] = s->cam.T[i];
  }
}

/* Same rounds, recording what oneRound LEAVES in the solver after each round, untouched: _H with the
 * damping on its diagonal (picp_solver.cpp:102), _b, (chi_in, chi_out, n_in), pose.  For bitwise
 * comparisons (the trace above removes the damping again, which rounds). */
void FN(picp_solve_raw)(FN(picp) * s, const int *corr, int n, int keep_outliers, int n_iters,
                        REAL *trace_H, REAL *trace_b, REAL *trace_stats, REAL *trace_T) {
  for (int it = 0; it < n_iters; ++it) {
    FN(picp_one_round)(s, corr, n, keep_outliers);
    for (int i = 0; i < 36; ++i) trace_H[36 * it + i] = s->H[i];
    for (int i = 0; i < 6; ++i) trace_b[6 * it + i] = s->b[i];
    trace_stats[3 * it + 0] = s->chi_inliers;
    trace_stats[3 * it + 1] = s->chi_outliers;
    trace_stats[3 * it + 2] = (REAL)s->num_inliers;
    for (int i = 0; i < 16; ++i) trace_T[16 * it + i] = s->cam.T[i];
  }
}

/* ---- small inverses used by triangulate_points ------------------------- */
/* Isometry inverse (Eigen Transform::inverse(Isometry)): R^T, -R^T t */
void FN(iso_inverse)(const REAL X[16], REAL iX[16]) {
  REAL out[16];
  for (int i = 0; i < 16; ++i) out[i] = (REAL)0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out[r + 4 * c] = X[c + 4 * r];
  REAL rt[3];
  FN(mat3_vec)(out, 4, X + 12, rt);
  for (int i = 0; i < 3; ++i) out[12 + i] = -rt[i];
  out[15] = (REAL)1;
  for (int i = 0; i < 16; ++i) iX[i] = out[i];
}

/* Matrix3f::inverse() (Eigen LU/InverseImpl.h, compute_inverse<.,.,3>):
 * cofactors, det from the first column, multiply by 1/det. */
void FN(mat3_inverse)(const REAL m[9], REAL inv[9]) {
#define A(r, c) m[(r) + 3 * (c)]
#define COF(i1, i2, j1, j2) (A(i1, j1) * A(i2, j2) - A(i1, j2) * A(i2, j1))
  /* cofactor(i,j) with cyclic indices, as Eigen's cofactor_3x3 */
  REAL c00 = COF(1, 2, 1, 2), c10 = COF(2, 0, 1, 2), c20 = COF(0, 1, 1, 2);
  const REAL det = c00 * A(0, 0) + (c10 * A(1, 0) + c20 * A(2, 0));
  const REAL invdet = (REAL)1 / det;
  REAL c01 = COF(1, 2, 2, 0), c11 = COF(2, 0, 2, 0), c21 = COF(0, 1, 2, 0);
  REAL c02 = COF(1, 2, 0, 1), c12 = COF(2, 0, 0, 1), c22 = COF(0, 1, 0, 1);
  /* inverse(r,c) = cofactor(c,r) / det */
  inv[0 + 3 * 0] = c00 * invdet; inv[0 + 3 * 1] = c10 * invdet; inv[0 + 3 * 2] = c20 * invdet;
  inv[1 + 3 * 0] = c01 * invdet; inv[1 + 3 * 1] = c11 * invdet; inv[1 + 3 * 2] = c21 * invdet;
  inv[2 + 3 * 0] = c02 * invdet; inv[2 + 3 * 1] = c12 * invdet; inv[2 + 3 * 2] = c22 * invdet;
#undef COF
#undef A
}

/* ---- triangulate_point, utils.cpp:36-49 -------------------------------- */
int FN(triangulate_point)(const REAL d1[3], const REAL d2[3], const REAL p2[3], REAL p[3]) {
  /* D = [-d1 d2] (3x2); DtD 2x2; Dtp 2x1 : :37-40 */
  const REAL D0[3] = {-d1[0], -d1[1], -d1[2]};
  REAL DtD[4], Dtp[2], ss[2];
  DtD[0] = FN(dot3)(D0[0], D0[0], D0[1], D0[1], D0[2], D0[2]);
  DtD[1] = FN(dot3)(d2[0], D0[0], d2[1], D0[1], d2[2], D0[2]);
  DtD[2] = FN(dot3)(D0[0], d2[0], D0[1], d2[1], D0[2], d2[2]);
  DtD[3] = FN(dot3)(d2[0], d2[0], d2[1], d2[1], d2[2], d2[2]);
  Dtp[0] = FN(dot3)(D0[0], p2[0], D0[1], p2[1], D0[2], p2[2]);
  Dtp[1] = FN(dot3)(d2[0], p2[0], d2[1], p2[1], d2[2], p2[2]);
  FN(ldlt_solve)(2, DtD, Dtp, ss);
  ss[0] = -ss[0]; ss[1] = -ss[1];
  if (ss[0] < (REAL)0 || ss[1] < (REAL)0) return 0;               /* :41 */
  for (int i = 0; i < 3; ++i) {
    const REAL a = ss[0] * d1[i];                                 /* :44 */
    const REAL b = p2[i] + ss[1] * d2[i];                         /* :45 */
    p[i] = (REAL)0.5 * (a + b);                                   /* :47 */
  }
  return 1;
}

/* ---- triangulate_points v1/v2/v3, utils.cpp:51-134 --------------------- */
/* out_pairs / app2 / out_app may be null (v1: points only; v2: + pairs;
 * v3: + appearance of the second image's point, utils.cpp:127). */
int FN(triangulate_points)(const REAL K[9], const REAL X[16], const int *corr, int n,
                           const REAL *p1, const REAL *p2, const REAL *app2, REAL *out_xyz,
                           int *out_pairs, REAL *out_app) {
  REAL iX[16], iK[9], iRiK[9];
  FN(iso_inverse)(X, iX);                                         /* :79 */
  FN(mat3_inverse)(K, iK);                                        /* :80 */
  FN(mat3_mul)(iX, 4, iK, 3, iRiK, 3);                            /* :81 */
  const REAL *t = iX + 12;                                        /* :82 */
  int n_success = 0;
  for (int k = 0; k < n; ++k) {
    const int i1 = corr[2 * k], i2 = corr[2 * k + 1];
    const REAL h1[3] = {p1[2 * i1], p1[2 * i1 + 1], (REAL)1};
    const REAL h2[3] = {p2[2 * i2], p2[2 * i2 + 1], (REAL)1};
    REAL d1[3], d2[3], p[3];
    FN(mat3_vec)(iK, 3, h1, d1);                                  /* :89-91 */
    FN(mat3_vec)(iRiK, 3, h2, d2);                                /* :92-94 */
    if (FN(triangulate_point)(d1, d2, t, p)) {
      if (out_pairs) { out_pairs[2 * n_success] = i2; out_pairs[2 * n_success + 1] = n_success; } /* :97 */
      out_xyz[3 * n_success] = p[0]; out_xyz[3 * n_success + 1] = p[1]; out_xyz[3 * n_success + 2] = p[2];
      if (out_app && app2)
        for (int a = 0; a < 10; ++a) out_app[10 * n_success + a] = app2[10 * i2 + a];
      n_success++;
    }
  }
  return n_success;
}

/* ---- compute_correspondences_images, vo_complete.cpp:12-49 ------------- */
/* Semantics of TreeNode_::bestMatchFull (eigen_kdtree.h:90-115) +
 * bruteForceBestMatch (brute_force_search.h:22-41): exact nearest neighbour
 * among points with squared distance < radius*radius (strict).  The tree's
 * tie order is an artefact of its PCA partition; the oracle scans in index
 * order with strict '<', i.e. the lowest index wins exact ties. */
static inline REAL FN(sqdist10)(const REAL *a, const REAL *b) {
  REAL s = (REAL)0;
  for (int k = 0; k < 10; ++k) { const REAL d = a[k] - b[k]; s += d * d; }
  return s;
}

int FN(match)(const REAL *a1, int n1, const REAL *a2, int n2, REAL radius, int *out_pairs) {
  const int tree_is_1 = (n1 >= n2);                               /* :15,:20 ties -> a1 */
  const REAL *tree = tree_is_1 ? a1 : a2;
  const REAL *qry = tree_is_1 ? a2 : a1;
  const int nt = tree_is_1 ? n1 : n2, nq = tree_is_1 ? n2 : n1;
  const REAL r2 = radius * radius;                                /* brute_force_search.h:31 */
  int n_out = 0;
  for (int q = 0; q < nq; ++q) {                                  /* :37 */
    int best = -1;
    REAL best_d = r2;
    for (int j = 0; j < nt; ++j) {
      const REAL d = FN(sqdist10)(tree + 10 * j, qry + 10 * q);
      if (d < best_d) { best_d = d; best = j; }                   /* brute_force_search.h:35 */
    }
    if (best >= 0) {
      if (tree_is_1) { out_pairs[2 * n_out] = best; out_pairs[2 * n_out + 1] = q; }   /* :41 */
      else { out_pairs[2 * n_out] = q; out_pairs[2 * n_out + 1] = best; }             /* :43 */
      n_out++;
    }
  }
  return n_out;
}

/* ---- rigid transform of a point set, PointCloud.h:77-82 ---------------- */
void FN(transform_points)(const REAL T[16], const REAL *in, int n, REAL *out) {
  for (int i = 0; i < n; ++i) {
    REAL p[3];
    FN(iso_point)(T, in + 3 * i, p);
    out[3 * i] = p[0]; out[3 * i + 1] = p[1]; out[3 * i + 2] = p[2];
  }
}

#undef FN
#undef CAT
#undef CAT_
