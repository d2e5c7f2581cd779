// vo_math.h -- small fixed-size math shared by the HIP kernels and the host
// side of libvo_hip.so.  Everything is float32, column-major like the Eigen
// objects of the reference (defs.h:7-29).
//
// The library is compiled with -ffp-contract=off: everything that must round
// like the reference's SSE2 build (CMakeLists.txt:6-7, no FMA) -- the matcher's
// distances, the projection / transform / triangulation operators and their
// gates, the reference-order solver mode (picp_term_exact) -- is written in the
// reference's operation order and stays unfused.  Fused multiply-adds appear
// only where they are requested explicitly (vo_fma): in the DEFAULT solver
// mode's linearisation (picp_accumulate_t), whose sums are tree-reduced in
// another order than the reference's sequential loop anyway, and in its 6x6
// solve.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VO_HD __host__ __device__ __forceinline__
#else
#define VO_HD inline
#endif

namespace vo {

VO_HD float vo_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// 1/z of the default-mode linearisation: hardware reciprocal + one Newton step + the special-case fixup (0, inf, nan
// as the IEEE quotient gives them) -- 4 instructions for the 10 of the correctly rounded quotient, at most 1 ulp from it
VO_HD float vo_recip_z(float z) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float r = __builtin_amdgcn_rcpf(z);
  const float e = __builtin_fmaf(-z, r, 1.f);
  return __builtin_amdgcn_div_fixupf(__builtin_fmaf(r, e, r), z, 1.f);
#else
  return 1.0f / z;
#endif
}

// x * y with 0 * anything = 0, NaN and infinity included (v_mul_legacy_f32; _n: x * (-y)).  The default-mode linearisation zeroes
// the reciprocal depth of a correspondence that must not contribute; with this product everything derived from it is an exact
// zero whatever the rejected projection left in the other factor -- four selects instead of eight per correspondence.
VO_HD float vo_mul0(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return (a == 0.f || b == 0.f) ? 0.f : a * b;
#endif
}
VO_HD float vo_mul0_n(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_mul_legacy_f32_e64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return (a == 0.f || b == 0.f) ? 0.f : a * (-b);
#endif
}

// A correspondence whose index lies outside its point array is DROPPED by the gather kernels: its world x carries this
// bit pattern (a quiet NaN with a payload no arithmetic produces) and the linearisation skips it.  A world point that
// really is NaN is not such a marker: it goes through the arithmetic like in the reference (and poisons the pose, like
// in the reference).
constexpr uint32_t VO_DROPPED_BITS = 0x7fc0d0d0u;
VO_HD bool is_dropped(float wx) { return __builtin_bit_cast(uint32_t, wx) == VO_DROPPED_BITS; }

// Eigen 3.4 sums a fixed-size, non-vectorised 3-term inner product as
// x0 + (x1 + x2) (Core/Redux.h, redux_novec_unroller).
VO_HD float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  return a0 * b0 + (a1 * b1 + a2 * b2);
}

// y = M * x, M 3x3 col-major with leading dimension ld
VO_HD void mat3_vec(const float* M, int ld, const float x[3], float y[3]) {
  for (int i = 0; i < 3; ++i) y[i] = dot3(M[i], x[0], M[i + ld], x[1], M[i + 2 * ld], x[2]);
}

VO_HD void mat3_mul(const float* A, int la, const float* B, int lb, float* C, int lc) {
  float tmp[9];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r)
      tmp[r + 3 * c] = dot3(A[r], B[lb * c], A[r + la], B[1 + lb * c], A[r + 2 * la], B[2 + lb * c]);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) C[r + lc * c] = tmp[r + 3 * c];
}

// Rigid pose as the kernels hold it: R col-major 3x3 + t.
struct Pose {
  float R[9];
  float t[3];
};

VO_HD Pose pose_from_T16(const float T[16]) {
  Pose p;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) p.R[r + 3 * c] = T[r + 4 * c];
  p.t[0] = T[12]; p.t[1] = T[13]; p.t[2] = T[14];
  return p;
}

VO_HD void pose_to_T16(const Pose& p, float T[16]) {
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) T[r + 4 * c] = p.R[r + 3 * c];
    T[3 + 4 * c] = 0.f;
  }
  T[12] = p.t[0]; T[13] = p.t[1]; T[14] = p.t[2]; T[15] = 1.f;
}

// Isometry3f * Vector3f: res = t; res += R * p   (camera.h:27, PointCloud.h:80)
VO_HD void pose_apply(const Pose& X, float px, float py, float pz, float& ox, float& oy, float& oz) {
  ox = X.t[0] + dot3(X.R[0], px, X.R[3], py, X.R[6], pz);
  oy = X.t[1] + dot3(X.R[1], px, X.R[4], py, X.R[7], pz);
  oz = X.t[2] + dot3(X.R[2], px, X.R[5], py, X.R[8], pz);
}

// Isometry * Isometry: R = Ra*Rb, t = Ra*tb + ta   (picp_solver.cpp:110)
VO_HD Pose pose_mul(const Pose& A, const Pose& B) {
  Pose C;
  mat3_mul(A.R, 3, B.R, 3, C.R, 3);
  float rt[3];
  mat3_vec(A.R, 3, B.t, rt);
  for (int i = 0; i < 3; ++i) C.t[i] = rt[i] + A.t[i];
  return C;
}

// Isometry inverse: R^T, -(R^T t)   (utils.cpp:79)
VO_HD Pose pose_inverse(const Pose& X) {
  Pose I;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) I.R[r + 3 * c] = X.R[c + 3 * r];
  float rt[3];
  mat3_vec(I.R, 3, X.t, rt);
  for (int i = 0; i < 3; ++i) I.t[i] = -rt[i];
  return I;
}

// Matrix3f::inverse(): cofactors over the determinant (utils.cpp:80)
VO_HD void mat3_inverse(const float m[9], float inv[9]) {
#define VO_A(r, c) m[(r) + 3 * (c)]
#define VO_COF(i1, i2, j1, j2) (VO_A(i1, j1) * VO_A(i2, j2) - VO_A(i1, j2) * VO_A(i2, j1))
  const float c00 = VO_COF(1, 2, 1, 2), c10 = VO_COF(2, 0, 1, 2), c20 = VO_COF(0, 1, 1, 2);
  const float det = c00 * VO_A(0, 0) + (c10 * VO_A(1, 0) + c20 * VO_A(2, 0));
  const float invdet = 1.f / det;
  const float c01 = VO_COF(1, 2, 2, 0), c11 = VO_COF(2, 0, 2, 0), c21 = VO_COF(0, 1, 2, 0);
  const float c02 = VO_COF(1, 2, 0, 1), c12 = VO_COF(2, 0, 0, 1), c22 = VO_COF(0, 1, 0, 1);
  inv[0] = c00 * invdet; inv[3] = c10 * invdet; inv[6] = c20 * invdet;
  inv[1] = c01 * invdet; inv[4] = c11 * invdet; inv[7] = c21 * invdet;
  inv[2] = c02 * invdet; inv[5] = c12 * invdet; inv[8] = c22 * invdet;
#undef VO_COF
#undef VO_A
}

// ---- camera ---------------------------------------------------------------
struct CamK {
  float K[9];
  int rows, cols, z_near, z_far;
};

// Camera::projectPoint (camera.h:25-37).  pc = camera-frame point, ph = K*pc.
// `inv` is the reciprocal the reference takes in double and rounds to float:
// identical to the correctly rounded float division (53 >= 2*24+2).
VO_HD bool project_point(const CamK& cam, const Pose& T, float px, float py, float pz, float& u,
                         float& v, float pc[3], float ph[3], float& inv) {
  pose_apply(T, px, py, pz, pc[0], pc[1], pc[2]);
  const bool z_ok = !(pc[2] > (float)cam.z_far || pc[2] < (float)cam.z_near);
  mat3_vec(cam.K, 3, pc, ph);
  inv = 1.0f / ph[2];
  u = ph[0] * inv;
  v = ph[1] * inv;
  const bool in_img = !(u < 0.f || u > (float)(cam.cols - 1)) && !(v < 0.f || v > (float)(cam.rows - 1));
  return z_ok && in_img;
}

// ---- PICP per-correspondence term ----------------------------------------
// Accumulator layout (NACC floats): 0..20 upper triangle of H row by row
// ((0,0),(0,1)..(0,5),(1,1)..(5,5)), 21..26 b, 27 chi_inliers, 28 chi_outliers,
// 29 number of inliers.
constexpr int NACC = 30;

// True iff K = [fx 0 cx; 0 fy cy; 0 0 1] exactly.  For such a K every product
// with a structural 0/1 of K is exact (0*x = 0, x+0 = x, x*1 = x), so the
// PINHOLE instantiations below drop those terms and still round bit for bit
// like the general 3x3 formulas.
VO_HD bool is_pinhole(const float K[9]) {
  return K[1] == 0.f && K[2] == 0.f && K[3] == 0.f && K[5] == 0.f && K[8] == 1.f;
}

// errorAndJacobian + the body of linearize's loop (picp_solver.cpp:25-53,
// :62-95) for one correspondence: world point w, measurement z.
// Written without divergent control flow: the reference's "continue"/"if"
// decisions become predicates, and a term that must not contribute has the
// inputs of its Jacobian zeroed so that no inf/nan of a rejected projection
// can reach an accumulator.  A world x of VO_DROPPED_BITS marks a dropped correspondence.
//
// This is the DEFAULT mode's arithmetic: the reference's formulas with one FMA per product and the reciprocal of
// vo_recip_z -- every intermediate within a few ulp of the reference's unfused value, so a decision (depth / image
// gate, chi^2 test) can differ from the reference's only for a correspondence that lies within that rounding of the
// gate itself: measured on 49 000 correspondences planted at every gate (tests/test_gpu_gates.py) <= 2 ulp of the gate for
// the depth gates, <= 3.7 for the image gates, <= 276 ulp of thr for chi^2 (= e0^2 + e1^2: the pixel error of (u, v) times
// 2 |e|, 3e-5 relative at thr = 100); the test holds the bands 4 / 8 / 1024.  Taking the decisions on reference-order
// values costs the batched solver 20 % (126 against 99 instructions per correspondence); a guard that
// re-evaluates only near-gate correspondences was measured too -- its detection arithmetic alone (commit "Decisions at the gates under rounding", -DVO_GUARD_PROBE)
// costs 9 % -- and not kept; sums: H_rc += J0r*J0c ; H_rc += J1r*J1c (lambda, when it is not 1, folded into the left factor)
// instead of (J0r*J0c + J1r*J1c)*lambda followed by an add.  The batched solver is bound by VALU issue, and this form
// is ~20 instructions per correspondence shorter than the unfused one.  Reference-order arithmetic, decisions
// included, is picp_term_exact below (vo_picp_set_exact).
// MUL0: the zeroing of a term that must not contribute through vo_mul0 (four selects instead of eight; same values for every
// contributing term).  The batched solver, bound by VALU issue, takes it (-2 %: 1.56 -> 1.53 ms per 200 x 50k x 50 rounds); the
// launch-per-round kernels, a latency chain, do not (the VOP3 products lengthen it: 4.57 -> 4.60 us per round).
template <bool PINHOLE, bool KEEP, bool STATS = true, bool MUL0 = false>
VO_HD void picp_accumulate_t(const CamK& cam, const Pose& T, float thr, float wx, float wy, float wz,
                             float zu, float zv, float acc[NACC]) {
  constexpr bool keep_outliers = KEEP;
  // pc = t + R p (camera.h:27), one FMA per product
  const float pc0 = vo_fma(T.R[6], wz, vo_fma(T.R[3], wy, vo_fma(T.R[0], wx, T.t[0])));
  const float pc1 = vo_fma(T.R[7], wz, vo_fma(T.R[4], wy, vo_fma(T.R[1], wx, T.t[1])));
  const float pc2 = vo_fma(T.R[8], wz, vo_fma(T.R[5], wy, vo_fma(T.R[2], wx, T.t[2])));
  const bool z_ok = !(pc2 > (float)cam.z_far || pc2 < (float)cam.z_near);       // camera.h:28
  float ph0, ph1, ph2;                                                          // camera.h:30
  if (PINHOLE) {
    ph0 = vo_fma(cam.K[0], pc0, cam.K[6] * pc2);
    ph1 = vo_fma(cam.K[4], pc1, cam.K[7] * pc2);
    ph2 = pc2;
  } else {
    ph0 = vo_fma(cam.K[0], pc0, vo_fma(cam.K[3], pc1, cam.K[6] * pc2));
    ph1 = vo_fma(cam.K[1], pc0, vo_fma(cam.K[4], pc1, cam.K[7] * pc2));
    ph2 = vo_fma(cam.K[2], pc0, vo_fma(cam.K[5], pc1, cam.K[8] * pc2));
  }
  float iz = vo_recip_z(ph2);                                                   // camera.h:31, picp_solver.cpp:44
  const float u = ph0 * iz, v = ph1 * iz;
  const bool in_img = !(u < 0.f || u > (float)(cam.cols - 1)) && !(v < 0.f || v > (float)(cam.rows - 1));
  const bool ok = z_ok && in_img && !is_dropped(wx);                            // :32-34, :72-73
  float e0 = u - zu, e1 = v - zv;                                               // :35
  const float chi = vo_fma(e0, e0, e1 * e1);                                    // :75
  const bool inl = ok && !(chi > thr);                                          // :78 (strict >)
  const bool outl = ok && (chi > thr);
  if (STATS) {                                    // (a caller that reports no statistics for this round leaves them out)
    acc[27] += inl ? chi : 0.f;                                                 // :86
    acc[28] += outl ? chi : 0.f;                                                // :82
    acc[29] += inl ? 1.f : 0.f;                                                 // :87
  }
  float lambda = inl ? 1.f : 0.f;
  if (keep_outliers) lambda = outl ? sqrtf(thr / chi) : lambda;                 // :80, :90
  const bool use = lambda != 0.f;
  // Jp*K (:39-51): row 0 = iz * (K row 0 - u * K row 2), row 1 = iz * (K row 1 - v * K row 2) -- the reference's
  // iz*K_0c + (-ph0*iz^2)*K_2c with u = ph0*iz taken out.  A term that must not contribute has the generators of its
  // Jacobian zeroed, so that no inf/nan of a rejected projection reaches an accumulator.  MUL0: it gets iz = 0, and every
  // product that could meet an inf / NaN is taken with vo_mul0 (0 * anything = 0): all its Jacobian entries are exact zeros;
  // what still needs a select are the residuals (they multiply the zero entries in plain FMAs) and the coordinates that
  // enter an FMA as a factor.
  iz = use ? iz : 0.f;
  e0 = use ? e0 : 0.f;
  e1 = use ? e1 : 0.f;
  const float p0 = (MUL0 || use) ? pc0 : 0.f, p2 = use ? pc2 : 0.f;
  const float un = (MUL0 || use) ? -u : 0.f, vn = (MUL0 || use) ? -v : 0.f;
  auto mul = [](float a, float b) { return MUL0 ? vo_mul0(a, b) : a * b; };
  auto mul_n = [](float a, float b) { return MUL0 ? vo_mul0_n(a, b) : a * (-b); };
  float J0[6], J1[6];
  if (PINHOLE) {
    const float p1 = (MUL0 || use) ? pc1 : 0.f;
    const float a = iz * cam.K[0], b = iz * cam.K[4];
    const float c0 = mul(iz, cam.K[6] + un), c1 = mul(iz, cam.K[7] + vn);
    J0[0] = a;   J0[1] = 0.f; J0[2] = c0;
    J1[0] = 0.f; J1[1] = b;   J1[2] = c1;
    // J = (Jp K)[I | skew(-pc)]
    J0[3] = mul(c0, p1);
    J1[3] = vo_fma(b, -p2, mul(c1, p1));
    J0[4] = vo_fma(a, p2, mul_n(c0, p0));
    J1[4] = mul_n(c1, p0);
    J0[5] = mul_n(a, p1);
    J1[5] = mul(b, p0);
  } else {
    // (the same roundings as the pinhole form for a pinhole K: fma(x, 0, k) = k, fma(x, 1, k) = k + x)
    const float p1 = use ? pc1 : 0.f;
    float A0[3], A1[3];
    for (int c = 0; c < 3; ++c) {
      A0[c] = mul(iz, vo_fma(un, cam.K[2 + 3 * c], cam.K[3 * c]));
      A1[c] = mul(iz, vo_fma(vn, cam.K[2 + 3 * c], cam.K[1 + 3 * c]));
    }
    // skew(v) = [0 -v2 v1; v2 0 -v0; -v1 v0 0] with v = -pc  (utils.h:96-102)
    const float v1 = -p1, v2 = -p2;
    J0[0] = A0[0]; J0[1] = A0[1]; J0[2] = A0[2];
    J1[0] = A1[0]; J1[1] = A1[1]; J1[2] = A1[2];
    J0[3] = vo_fma(A0[1], v2, mul(A0[2], p1));
    J1[3] = vo_fma(A1[1], v2, mul(A1[2], p1));
    J0[4] = vo_fma(A0[0], -v2, mul_n(A0[2], p0));
    J1[4] = vo_fma(A1[0], -v2, mul_n(A1[2], p0));
    J0[5] = vo_fma(A0[0], v1, mul(A0[1], p0));
    J1[5] = vo_fma(A1[0], v1, mul(A1[1], p0));
  }
  // left factors, scaled by lambda only when it can differ from 1
  float L0[6], L1[6];
  for (int r = 0; r < 6; ++r) {
    L0[r] = keep_outliers ? J0[r] * lambda : J0[r];
    L1[r] = keep_outliers ? J1[r] * lambda : J1[r];
  }
  // H += J^T J * lambda ; b += J^T e * lambda                 (:92-93)
  int k = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) {
      // structural zeros of the pinhole Jacobian: J0[1] = 0, J1[0] = 0
      const bool z0 = PINHOLE && (r == 1 || c == 1);
      const bool z1 = PINHOLE && (r == 0 || c == 0);
      if (!z0) acc[k] = vo_fma(L0[r], J0[c], acc[k]);
      if (!z1) acc[k] = vo_fma(L1[r], J1[c], acc[k]);
      ++k;
    }
  for (int r = 0; r < 6; ++r) {
    const bool z0 = PINHOLE && r == 1, z1 = PINHOLE && r == 0;
    if (!z0) acc[21 + r] = vo_fma(L0[r], e0, acc[21 + r]);
    if (!z1) acc[21 + r] = vo_fma(L1[r], e1, acc[21 + r]);
  }
}

// ---- reference-order ("exact") form of the same term --------------------------
// What linearize adds for ONE correspondence, rounded exactly like the reference's
// scalar loop (picp_solver.cpp:62-95): J from the general 3x3 formulas (:39-51), every
// product unfused, H_rc term = (J0r*J0c + J1r*J1c)*lambda, b_r term = (J0r*e0 + J1r*e1)*lambda.
// term[0..20] upper triangle of J^T J * lambda row by row (the product is symmetric bit for
// bit: float multiplication commutes), term[21..26] J^T e * lambda, term[27] chi.
// Returns 0 = skipped before any statistic (:72-73), 1 = inlier, 2 = outlier (:78).
// The caller adds term k to its running sum IN CORRESPONDENCE ORDER: that sequential sum is
// what makes the result bit-identical to the reference arithmetic (picp_exact_kernel).
constexpr int NTERM = 28;
VO_HD int picp_term_exact(const CamK& cam, const Pose& T, float thr, float wx, float wy, float wz, float zu,
                          float zv, float term[NTERM]) {
  float pc[3], ph[3];
  pose_apply(T, wx, wy, wz, pc[0], pc[1], pc[2]);                       // camera.h:27
  if (is_dropped(wx)) return 0;                                         // dropped correspondence (pack marker)
  if (pc[2] > (float)cam.z_far || pc[2] < (float)cam.z_near) return 0;  // camera.h:28
  mat3_vec(cam.K, 3, pc, ph);                                           // camera.h:30
  const float iz = 1.0f / ph[2];                                        // camera.h:31, picp_solver.cpp:44
  const float u = ph[0] * iz, v = ph[1] * iz;
  if (u < 0.f || u > (float)(cam.cols - 1)) return 0;                   // camera.h:32
  if (v < 0.f || v > (float)(cam.rows - 1)) return 0;                   // camera.h:34
  const float e0 = u - zu, e1 = v - zv;                                 // :35
  const float chi = e0 * e0 + e1 * e1;                                  // :75
  const float iz2 = iz * iz;                                            // :45
  const float g0 = -ph[0] * iz2, g1 = -ph[1] * iz2;                     // :47-49
  float A0[3], A1[3];                                                   // Jp*K, :51 (Jp's structural zeros dropped)
  for (int c = 0; c < 3; ++c) {
    A0[c] = iz * cam.K[3 * c] + g0 * cam.K[2 + 3 * c];
    A1[c] = iz * cam.K[1 + 3 * c] + g1 * cam.K[2 + 3 * c];
  }
  const float v0 = -pc[0], v1 = -pc[1], v2 = -pc[2];                    // skew(-pc), utils.h:96-102
  float J0[6], J1[6];
  J0[0] = A0[0]; J0[1] = A0[1]; J0[2] = A0[2];
  J1[0] = A1[0]; J1[1] = A1[1]; J1[2] = A1[2];
  J0[3] = A0[1] * v2 + A0[2] * (-v1);
  J1[3] = A1[1] * v2 + A1[2] * (-v1);
  J0[4] = A0[0] * (-v2) + A0[2] * v0;
  J1[4] = A1[0] * (-v2) + A1[2] * v0;
  J0[5] = A0[0] * v1 + A0[1] * (-v0);
  J1[5] = A1[0] * v1 + A1[1] * (-v0);
  const bool outl = chi > thr;                                          // :78 (strict)
  const float lambda = outl ? sqrtf(thr / chi) : 1.f;                   // :80 (double sqrt rounded = float sqrt)
  int k = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) {
      const float jtj = J0[r] * J0[c] + J1[r] * J1[c];
      term[k++] = jtj * lambda;                                         // :92
    }
  for (int r = 0; r < 6; ++r) {
    const float jte = J0[r] * e0 + J1[r] * e1;
    term[21 + r] = jte * lambda;                                        // :93
  }
  term[27] = chi;
  return outl ? 2 : 1;
}

VO_HD void picp_accumulate(const CamK& cam, const Pose& T, float thr, bool keep_outliers, float wx,
                           float wy, float wz, float zu, float zv, float acc[NACC]) {
  if (keep_outliers) picp_accumulate_t<false, true>(cam, T, thr, wx, wy, wz, zu, zv, acc);
  else picp_accumulate_t<false, false>(cam, T, thr, wx, wy, wz, zu, zv, acc);
}

// ---- Eigen::LDLT (pivoted, lower) for the 6x6 normal equations ------------
// Restates ldlt_inplace<Lower>::unblocked + LDLT::_solve_impl
// (picp_solver.cpp:109).  `a` is the full symmetric matrix (row r, col c at
// a[r][c]); every index below is a compile-time constant after unrolling so
// the matrix lives in registers; the data-dependent pivot is handled by
// comparing against constants.
VO_HD void sym_swap6(float a[6][6], int k, int p) {
#pragma unroll
  for (int j = 0; j < 6; ++j) { const float t = a[k][j]; a[k][j] = a[p][j]; a[p][j] = t; }
#pragma unroll
  for (int i = 0; i < 6; ++i) { const float t = a[i][k]; a[i][k] = a[i][p]; a[i][p] = t; }
}

VO_HD void ldlt6_solve(float a[6][6], const float rhs[6], float x[6]) {
  int tr[6];
  bool zero = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (zero) { tr[k] = k; continue; }
    int piv = k;
    float big = fabsf(a[k][k]);
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const float v = fabsf(a[i][i]);
      if (v > big) { big = v; piv = i; }
    }
    tr[k] = piv;
#pragma unroll
    for (int p = k + 1; p < 6; ++p)
      if (piv == p) sym_swap6(a, k, p);
    if (k > 0) {
      float tmp[6];
#pragma unroll
      for (int j = 0; j < k; ++j) tmp[j] = a[j][j] * a[k][j];
      float accd = 0.f;
#pragma unroll
      for (int j = 0; j < k; ++j) accd += a[k][j] * tmp[j];
      a[k][k] -= accd;
#pragma unroll
      for (int i = k + 1; i < 6; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < k; ++j) s += a[i][j] * tmp[j];
        a[i][k] -= s;
      }
    }
    const float akk = a[k][k];
    const bool ok = fabsf(akk) > 0.f;
    if (k == 0 && !ok) { zero = true; tr[0] = 0; continue; }
    if (ok) {
#pragma unroll
      for (int i = k + 1; i < 6; ++i) a[i][k] /= akk;
    }
  }
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] = rhs[i];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int p = k + 1; p < 6; ++p)
      if (tr[k] == p) { const float t = y[k]; y[k] = y[p]; y[p] = t; }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = y[i];
#pragma unroll
    for (int j = 0; j < i; ++j) s -= a[i][j] * y[j];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] = (fabsf(a[i][i]) > 1.17549435e-38f) ? y[i] / a[i][i] : 0.f;
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float s = y[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) s -= a[j][i] * y[j];
    y[i] = s;
  }
#pragma unroll
  for (int k = 5; k >= 0; --k) {
#pragma unroll
    for (int p = k + 1; p < 6; ++p)
      if (tr[k] == p) { const float t = y[k]; y[k] = y[p]; y[p] = t; }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// ---- latency-optimised form of the same factorisation ------------------------
// Eigen's unblocked LDLT is left-looking: at step k it has updated only the
// columns < k, so the "largest remaining diagonal" it pivots on is always an
// ORIGINAL diagonal entry.  The whole pivot sequence therefore follows from the
// six original diagonal values alone (simulated below exactly as Eigen swaps
// them, ties included), and applying that permutation up front turns the
// factorisation into the swap-free algorithm on P A P^T -- the same arithmetic
// in the same order as ldlt6_solve, without 15 data-dependent symmetric swaps.
// The permuted entries are gathered from memory (LDS on the device) with
// computed addresses; everything after that uses compile-time indices.
// Device build: one Newton-refined reciprocal per column replaces Eigen's
// per-element divisions and the updates use FMA (<= ~1 ulp per entry).
VO_HD float vo_recip(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(x);
  r = vo_fma(vo_fma(-x, r, 1.0f), r, r);
  return r;
#else
  return 1.0f / x;
#endif
}

// full36: symmetric 6x6 (row r, col c at full36[6*r+c], damping included);
// rhs6: right-hand side; scratch6: 6 writable floats in the same memory space.
VO_HD void ldlt6_solve_perm(const float* full36, const float* rhs6, float* scratch6, float x[6]) {
  float dv[6];
  int org[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { dv[i] = fabsf(full36[7 * i]); org[i] = i; }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int piv = k;
    float big = dv[k];
#pragma unroll
    for (int i = k + 1; i < 6; ++i) {
      const bool g = dv[i] > big;       // strict: first maximum wins, as maxCoeff
      big = g ? dv[i] : big;
      piv = g ? i : piv;
    }
#pragma unroll
    for (int p = k + 1; p < 6; ++p) {
      const bool sel = piv == p;
      const float tv = dv[k]; dv[k] = sel ? dv[p] : dv[k]; dv[p] = sel ? tv : dv[p];
      const int to = org[k]; org[k] = sel ? org[p] : org[k]; org[p] = sel ? to : org[p];
    }
  }
  // B = P A P^T (lower triangle), y = P rhs
  float B[6][6];
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    y[i] = rhs6[org[i]];
#pragma unroll
    for (int j = 0; j <= i; ++j) B[i][j] = full36[6 * org[i] + org[j]];
  }
  float inv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (k > 0) {
      float tmp[6];
#pragma unroll
      for (int j = 0; j < k; ++j) tmp[j] = B[j][j] * B[k][j];
      float accd = 0.f;
#pragma unroll
      for (int j = 0; j < k; ++j) accd = vo_fma(B[k][j], tmp[j], accd);
      B[k][k] -= accd;
#pragma unroll
      for (int i = k + 1; i < 6; ++i) {
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < k; ++j) sacc = vo_fma(B[i][j], tmp[j], sacc);
        B[i][k] -= sacc;
      }
    }
    const float akk = B[k][k];
    inv[k] = (fabsf(akk) > 1.17549435e-38f) ? vo_recip(akk) : 0.f;
#pragma unroll
    for (int i = k + 1; i < 6; ++i) B[i][k] *= inv[k];
  }
#pragma unroll
  for (int i = 1; i < 6; ++i) {
    float sacc = y[i];
#pragma unroll
    for (int j = 0; j < i; ++j) sacc = vo_fma(-B[i][j], y[j], sacc);
    y[i] = sacc;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] *= inv[i];
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    float sacc = y[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) sacc = vo_fma(-B[j][i], y[j], sacc);
    y[i] = sacc;
  }
  // x = P^T y
#pragma unroll
  for (int i = 0; i < 6; ++i) scratch6[org[i]] = y[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = scratch6[i];
}

// ---- the factorisation without pivoting (fast mode since round 3: picp.hip, picp_tail_direct) ----------------
// B: lower triangle of a symmetric matrix in the order it is to be eliminated (B[i][j], j <= i), y: right-hand side; on
// return y = solution.  The arithmetic of ldlt6_solve_perm after its gather, operation for operation: with B = P A P^T
// it is that function; the kernels call it on H itself (H is positive definite: any order is backward stable).
// (Measured and not kept: the bare v_rcp_f32 without its two Newton FMAs -- no change of the round time.)
VO_HD void ldlt6_solve_ordered(float B[6][6], float y[6]) {
  float inv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (k > 0) {
      float tmp[6];
#pragma unroll
      for (int j = 0; j < k; ++j) tmp[j] = B[j][j] * B[k][j];
      float accd = 0.f;
#pragma unroll
      for (int j = 0; j < k; ++j) accd = vo_fma(B[k][j], tmp[j], accd);
      B[k][k] -= accd;
#pragma unroll
      for (int i = k + 1; i < 6; ++i) {
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < k; ++j) sacc = vo_fma(B[i][j], tmp[j], sacc);
        B[i][k] -= sacc;
      }
    }
    const float akk = B[k][k];
    inv[k] = (fabsf(akk) > 1.17549435e-38f) ? vo_recip(akk) : 0.f;
#pragma unroll
    for (int i = k + 1; i < 6; ++i) B[i][k] *= inv[k];
  }
#pragma unroll
  for (int i = 1; i < 6; ++i) {
    float sacc = y[i];
#pragma unroll
    for (int j = 0; j < i; ++j) sacc = vo_fma(-B[i][j], y[j], sacc);
    y[i] = sacc;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] *= inv[i];
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    float sacc = y[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) sacc = vo_fma(-B[j][i], y[j], sacc);
    y[i] = sacc;
  }
}

// sin and cos of a small angle, |x| <= 0.5 (a Gauss-Newton step; the callers fall back to sincosf beyond): minimax
// polynomials on the reduced range of the classic single-precision kernels (Cephes sinf / cosf), <= 1 ulp there.
// sincosf's general path (argument reduction, quadrant logic) is a quarter of the round's tail.
VO_HD void sincos_small(float x, float& s, float& c) {
  const float z = x * x;
  float ps = vo_fma(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = vo_fma(ps, z, -1.6666654611e-1f);
  s = vo_fma(ps * z, x, x);
  float pc = vo_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = vo_fma(pc, z, 4.166664568298827e-2f);
  c = vo_fma(pc * z, z, vo_fma(-0.5f, z, 1.0f));
}

// R = Rx Ry Rz written out: the 3-term products of utils.h:64-78 with the
// structural zeros/ones of the three factors removed (x*1, x+0 and 0*x are
// exact), so the entries round exactly like the generic product.
VO_HD Pose v2t_from_sincos(const float v[6], float sx, float cx, float sy, float cy, float sz, float cz) {
  // Rx*Ry = [cy 0 sy; sx*sy cx -sx*cy; -cx*sy sx cx*cy]
  const float a10 = sx * sy, a12 = -sx * cy, a20 = -cx * sy, a22 = cx * cy;
  Pose P;
  P.R[0] = cy * cz;              P.R[3] = cy * (-sz);            P.R[6] = sy;
  P.R[1] = a10 * cz + cx * sz;   P.R[4] = a10 * (-sz) + cx * cz; P.R[7] = a12;
  P.R[2] = a20 * cz + sx * sz;   P.R[5] = a20 * (-sz) + sx * cz; P.R[8] = a22;
  P.t[0] = v[0]; P.t[1] = v[1]; P.t[2] = v[2];
  return P;
}

// 2x2 instance of the same algorithm (utils.cpp:40): solves [m00 m10; m10 m11] x = rhs.
VO_HD void ldlt2_solve(float m00, float m10, float m11, float r0, float r1, float& x0, float& x1) {
  const bool sw = fabsf(m11) > fabsf(m00);
  if (sw) { const float t = m00; m00 = m11; m11 = t; }
  if (!(fabsf(m00) > 0.f)) { x0 = 0.f; x1 = 0.f; return; }   // zero matrix: D^+ = 0
  m10 /= m00;
  const float tmp0 = m00 * m10;
  m11 -= m10 * tmp0;
  float y0 = sw ? r1 : r0, y1 = sw ? r0 : r1;
  y1 -= m10 * y0;
  y0 = (fabsf(m00) > 1.17549435e-38f) ? y0 / m00 : 0.f;
  y1 = (fabsf(m11) > 1.17549435e-38f) ? y1 / m11 : 0.f;
  y0 -= m10 * y1;
  x0 = sw ? y1 : y0;
  x1 = sw ? y0 : y1;
}

// v2tEuler (utils.h:64-78): t = v[0:3], R = Rx(v3) Ry(v4) Rz(v5).  The
// reference's sin/cos resolve to the double libm functions rounded to float:
// v2t_euler_exact does the same on host and device.
VO_HD Pose v2t_euler_exact(const float v[6]) {
  const float sx = (float)sin((double)v[3]), cx = (float)cos((double)v[3]);
  const float sy = (float)sin((double)v[4]), cy = (float)cos((double)v[4]);
  const float sz = (float)sin((double)v[5]), cz = (float)cos((double)v[5]);
  return v2t_from_sincos(v, sx, cx, sy, cy, sz, cz);
}

VO_HD Pose v2t_euler(const float v[6]) {
#if defined(__HIP_DEVICE_COMPILE__)
  // device: float sincos (ocml); differs from the double-then-round value by
  // at most an ulp of the matrix entry -- far inside the pose tolerance.
  float sx, cx, sy, cy, sz, cz;
  sincosf(v[3], &sx, &cx);
  sincosf(v[4], &sy, &cy);
  sincosf(v[5], &sz, &cz);
  return v2t_from_sincos(v, sx, cx, sy, cy, sz, cz);
#else
  return v2t_euler_exact(v);
#endif
}

// Tail of PICPSolver::oneRound (picp_solver.cpp:102-110): from the reduced
// accumulators build H (+damping), solve H dx = -b, T <- v2tEuler(dx) * T.
// H_out (36, col-major, damping included) and b_out are what the reference
// leaves in _H/_b.
template <bool EXACT>
VO_HD Pose picp_update_t(const float acc[NACC], float damping, const Pose& T, float* H_out, float* b_out) {
  float a[6][6];
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) { a[r][c] = acc[k]; a[c][r] = acc[k]; ++k; }
#pragma unroll
  for (int i = 0; i < 6; ++i) a[i][i] += 1.f * damping;
  if (H_out) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) H_out[r + 6 * c] = a[r][c];
  }
  float nb[6], dx[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { nb[i] = -acc[21 + i]; if (b_out) b_out[i] = acc[21 + i]; }
  ldlt6_solve(a, nb, dx);
  const Pose dT = EXACT ? v2t_euler_exact(dx) : v2t_euler(dx);
  return pose_mul(dT, T);
}

VO_HD Pose picp_update(const float acc[NACC], float damping, const Pose& T, float* H_out, float* b_out) {
  return picp_update_t<false>(acc, damping, T, H_out, b_out);
}

// ---- triangulate_point (utils.cpp:36-49) -----------------------------------
VO_HD bool triangulate_point(const float d1[3], const float d2[3], const float p2[3], float p[3]) {
  const float D0[3] = {-d1[0], -d1[1], -d1[2]};
  const float m00 = dot3(D0[0], D0[0], D0[1], D0[1], D0[2], D0[2]);
  const float m10 = dot3(d2[0], D0[0], d2[1], D0[1], d2[2], D0[2]);
  const float m11 = dot3(d2[0], d2[0], d2[1], d2[1], d2[2], d2[2]);
  const float r0 = dot3(D0[0], p2[0], D0[1], p2[1], D0[2], p2[2]);
  const float r1 = dot3(d2[0], p2[0], d2[1], p2[1], d2[2], p2[2]);
  float s0, s1;
  ldlt2_solve(m00, m10, m11, r0, r1, s0, s1);
  s0 = -s0; s1 = -s1;
  if (s0 < 0.f || s1 < 0.f) return false;                              // :41
  for (int i = 0; i < 3; ++i) {
    const float a = s0 * d1[i];
    const float b = p2[i] + s1 * d2[i];
    p[i] = 0.5f * (a + b);                                             // :44-47
  }
  return true;
}

// Per-frame constants of triangulate_points (utils.cpp:79-82)
struct TriConst {
  float iK[9];
  float iRiK[9];
  float t[3];
};

VO_HD TriConst tri_constants(const float K[9], const Pose& X) {
  TriConst c;
  const Pose iX = pose_inverse(X);
  mat3_inverse(K, c.iK);
  mat3_mul(iX.R, 3, c.iK, 3, c.iRiK, 3);
  c.t[0] = iX.t[0]; c.t[1] = iX.t[1]; c.t[2] = iX.t[2];
  return c;
}

}  // namespace vo
