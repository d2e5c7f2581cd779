// vo/epipolar.hpp -- epipolar initialisation (epipolar_utils.h / .cpp:48-65,
// 103-213): normalised 8-point fundamental, rank-2 projection, E = K^T F K,
// the two rotations of the essential matrix, and the cheirality vote over the
// four (R, +-t) candidates by counting successful triangulations -- the vote
// runs the GPU triangulation kernel (overload v1).
//
// Host code, once per sequence.  The reference's JacobiSVD calls are replaced
// by vo/linalg.hpp (double inside, rounded to float at the end): the null
// vector of the N x 9 system is the eigenvector of A^T A for the smallest
// eigenvalue.  Singular-vector signs are conventions of the SVD routine; the
// candidate set {R1,R2} x {+t,-t} does not depend on them, only the order in
// which equally-voted candidates would be tried does.
#pragma once

#include <cstdio>
#include <cstdlib>

#include "linalg.hpp"
#include "utils.hpp"

namespace vo {

using IsometryPair = std::pair<Isometry3f, Isometry3f>;   // defs.h:17

//! coordinates scaled to [-1,1] by the per-axis maximum (epipolar_utils.cpp:48-65)
inline Vector2fVector normalize(const Vector2fVector& p, Matrix3f& T) {
  Vector2fVector ret(p.size());
  float max_x = 0.f, max_y = 0.f;
  for (const auto& v : p) { if (v.x() > max_x) max_x = v.x(); if (v.y() > max_y) max_y = v.y(); }
  for (size_t i = 0; i < p.size(); i++) { ret[i][0] = p[i].x() / (max_x / 2.f) - 1.f; ret[i][1] = p[i].y() / (max_y / 2.f) - 1.f; }
  T = Matrix3f::FromRows(1.f / (max_x / 2.f), 0.f, -1.f, 0.f, 1.f / (max_y / 2.f), -1.f, 0.f, 0.f, 1.f);
  return ret;
}

//! back half of estimate_fundamental (epipolar_utils.cpp:127-143) from the 81 entries of A^T A and the two conditioning
//! matrices: null vector, rank-2 projection, un-normalisation
inline Matrix3f fundamental_from_normal_matrix(const std::vector<double>& AtA, const Matrix3f& T1, const Matrix3f& T2) {
  std::vector<double> ev, evec;
  linalg::jacobi_eigen_sym(9, AtA, ev, evec);
  linalg::Mat3d Fa;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fa.m[i][j] = evec[(size_t)(3 * i + j) * 9 + 0];   // :128-131
  linalg::Mat3d U, V; double s[3];
  linalg::svd3(Fa, U, s, V);
  linalg::Mat3d D = linalg::Mat3d::zero();
  D.m[0][0] = s[0]; D.m[1][1] = s[1];                        // rank 2, :135-139
  const linalg::Mat3d F = U * D * V.transpose();
  linalg::Mat3d t1, t2;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { t1.m[i][j] = T1(i, j); t2.m[i][j] = T2(i, j); }
  const linalg::Mat3d Fd = t1.transpose() * F * t2;          // :142
  Matrix3f out;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out(i, j) = (float)Fd.m[i][j];
  return out;
}

//! the conditioning matrix normalize() builds from an image's per-axis maxima (epipolar_utils.cpp:61-63)
inline Matrix3f conditioning_matrix(float max_x, float max_y) {
  return Matrix3f::FromRows(1.f / (max_x / 2.f), 0.f, -1.f, 0.f, 1.f / (max_y / 2.f), -1.f, 0.f, 0.f, 1.f);
}

//! epipolar_utils.cpp:103-144
inline Matrix3f estimate_fundamental(const IntPairVector& correspondences, const Vector2fVector& p1_img,
                                     const Vector2fVector& p2_img) {
  if (correspondences.size() < 8) {
    std::printf("Less than 8 points available to compute the fundamental matrix, aborting . . .\n");
    std::exit(-1);                                            // :105-108
  }
  Matrix3f T1, T2;
  const Vector2fVector p1n = normalize(p1_img, T1), p2n = normalize(p2_img, T2);
  std::vector<double> AtA(81, 0.0);
  for (const IntPair& c : correspondences) {
    const double d1[3] = {p1n[(size_t)c.first].x(), p1n[(size_t)c.first].y(), 1.0};
    const double d2[3] = {p2n[(size_t)c.second].x(), p2n[(size_t)c.second].y(), 1.0};
    double row[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) row[3 * i + j] = d1[i] * d2[j];     // :124-125
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) AtA[(size_t)i * 9 + j] += row[i] * row[j];
  }
  return fundamental_from_normal_matrix(AtA, T1, T2);
}

//! epipolar_utils.cpp:146-174
inline IsometryPair essential2transformPair(const Matrix3f& E) {
  linalg::Mat3d Ed, w = linalg::Mat3d::zero();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ed.m[i][j] = E(i, j);
  w.m[0][1] = -1; w.m[1][0] = 1; w.m[2][2] = 1;
  linalg::Mat3d U, V; double s[3];
  linalg::svd3(Ed, U, s, V);
  linalg::Mat3d R1 = V * w * U.transpose();
  // :154-159 -- the reference decomposes -E when det(R1) < 0.  -E = (-U) S V^T, so that second decomposition yields
  // exactly -R1 (and -R2 below); negating U says the same without depending on how an SVD routine signs the vector
  // of the (near-)zero singular value, which for an exactly rank-2 E is free.
  if (R1.det() < 0) {
    U = -U;
    R1 = V * w * U.transpose();
  }
  auto make = [&](const linalg::Mat3d& R) {
    Isometry3f X = Isometry3f::Identity();
    const linalg::Mat3d ts = R * Ed;                         // t_skew = R*E, :162-163
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) X(i, j) = (float)R.m[i][j];
    X(0, 3) = (float)ts.m[2][1]; X(1, 3) = (float)ts.m[0][2]; X(2, 3) = (float)ts.m[1][0];
    return X;
  };
  const linalg::Mat3d R2 = V * w.transpose() * U.transpose();   // :166
  return IsometryPair(make(R1), make(R2));
}

//! epipolar_utils.cpp:176-213 with the cheirality count abstracted: count_in_front(X) returns the number of
//! correspondences that triangulate successfully under the candidate X (the reference calls
//! triangulate_points v1 and uses its return value, :196-209).  Strict '>' keeps the first best candidate.
//! the four candidates of epipolar_utils.cpp:187-211 in the order the reference tries them: (R1, t), (R1, -t), (R2, t), (R2, -t)
inline void transform_candidates(const Matrix3f& k, const Matrix3f& F, Isometry3f X[4]) {
  linalg::Mat3d kd, Fd;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { kd.m[i][j] = k(i, j); Fd.m[i][j] = F(i, j); }
  const linalg::Mat3d Ed = kd.transpose() * Fd * kd;          // :180
  Matrix3f E;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E(i, j) = (float)Ed.m[i][j];
  const IsometryPair X12 = essential2transformPair(E);
  for (int cand = 0; cand < 4; ++cand) {
    X[cand] = cand < 2 ? X12.first : X12.second;
    if (cand & 1) { X[cand](0, 3) = -X[cand](0, 3); X[cand](1, 3) = -X[cand](1, 3); X[cand](2, 3) = -X[cand](2, 3); }
  }
}
//! the vote: strict '>' keeps the first best candidate, no candidate in front of anything leaves the identity (:184-211)
inline Isometry3f pick_candidate(const Isometry3f X[4], const int n_in_front_of[4]) {
  int n_in_front = 0;
  Isometry3f X_best = Isometry3f::Identity();
  for (int cand = 0; cand < 4; ++cand)
    if (n_in_front_of[cand] > n_in_front) { n_in_front = n_in_front_of[cand]; X_best = X[cand]; }
  return X_best;
}

template <class CountInFront>
inline Isometry3f estimate_transform_with(const Matrix3f k, const IntPairVector& correspondences,
                                          const Vector2fVector& p1_img, const Vector2fVector& p2_img,
                                          CountInFront&& count_in_front) {
  const Matrix3f F = estimate_fundamental(correspondences, p1_img, p2_img);
  Isometry3f X[4];
  transform_candidates(k, F, X);
  int n[4];
  for (int cand = 0; cand < 4; ++cand) n[cand] = count_in_front(X[cand]);      // :187-211
  return pick_candidate(X, n);
}

//! epipolar_utils.cpp:176-213 -- pose of the first camera in the frame of the second
inline Isometry3f estimate_transform(const Matrix3f k, const IntPairVector& correspondences,
                                     const Vector2fVector& p1_img, const Vector2fVector& p2_img) {
  Vector3fVector triang;
  return estimate_transform_with(k, correspondences, p1_img, p2_img, [&](const Isometry3f& X_test) {
    return triangulate_points(k, X_test, correspondences, p1_img, p2_img, triang);
  });
}

}  // namespace vo
