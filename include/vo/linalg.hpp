// vo/linalg.hpp -- the few dense routines the cold (once-per-sequence) host
// code needs in place of Eigen: symmetric Jacobi eigen-solver and a 3x3 SVD
// built on it.  Double precision internally; callers round to float.
#pragma once

#include <algorithm>
#include <cmath>
#include <vector>

namespace vo {
namespace linalg {

// Cyclic Jacobi for a symmetric n x n matrix (row-major, destroyed).  On
// return evals[i] ascending and evecs column i (evecs[r*n+i]) the matching
// unit eigenvector.
inline void jacobi_eigen_sym(int n, std::vector<double> a, std::vector<double>& evals, std::vector<double>& evecs) {
  std::vector<double> v((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < n; ++p) for (int q = 0; q < n; ++q) (p == q ? diag : off) += a[(size_t)p * n + q] * a[(size_t)p * n + q];
    if (off <= 1e-30 * (diag + 1e-300)) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double theta = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {     // A <- A J
          const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
          a[(size_t)k * n + p] = c * akp - s * akq;
          a[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {     // A <- J^T A
          const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
          a[(size_t)p * n + k] = c * apk - s * aqk;
          a[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {     // V <- V J
          const double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
          v[(size_t)k * n + p] = c * vkp - s * vkq;
          v[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  std::vector<int> order((size_t)n);
  for (int i = 0; i < n; ++i) order[(size_t)i] = i;
  std::sort(order.begin(), order.end(), [&](int x, int y) { return a[(size_t)x * n + x] < a[(size_t)y * n + y]; });
  evals.assign((size_t)n, 0.0);
  evecs.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) {
    evals[(size_t)i] = a[(size_t)order[(size_t)i] * n + order[(size_t)i]];
    for (int r = 0; r < n; ++r) evecs[(size_t)r * n + i] = v[(size_t)r * n + order[(size_t)i]];
  }
}

struct Mat3d {
  double m[3][3];
  static Mat3d zero() { Mat3d z; for (auto& r : z.m) for (double& x : r) x = 0; return z; }
  static Mat3d identity() { Mat3d z = zero(); z.m[0][0] = z.m[1][1] = z.m[2][2] = 1; return z; }
  Mat3d operator*(const Mat3d& b) const {
    Mat3d c = zero();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) c.m[i][j] += m[i][k] * b.m[k][j];
    return c;
  }
  Mat3d transpose() const { Mat3d t; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t.m[i][j] = m[j][i]; return t; }
  Mat3d operator-() const { Mat3d t; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t.m[i][j] = -m[i][j]; return t; }
  double det() const {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
  }
};

// M = U diag(s) V^T, s descending, U and V orthogonal (full 3x3).
// One-sided (Hestenes) Jacobi: plane rotations applied from the right make the columns of A = M V mutually orthogonal;
// their norms are the singular values, the normalised columns the left singular vectors.  Unlike the eigen-decomposition of
// M^T M this keeps small singular values and their vectors accurate RELATIVE to their own size -- the essential matrix has
// one (near-)zero singular value, and its left singular vector enters the rotation candidates (epipolar_utils.cpp:154).
// Columns whose norm vanishes against the largest get their left vector by orthogonal completion.
inline void svd3(const Mat3d& M, Mat3d& U, double s[3], Mat3d& V) {
  double A[3][3], W[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = M.m[i][j];
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double app = 0, aqq = 0, apq = 0;
        for (int i = 0; i < 3; ++i) { app += A[i][p] * A[i][p]; aqq += A[i][q] * A[i][q]; apq += A[i][p] * A[i][q]; }
        if (std::fabs(apq) <= 1e-17 * std::sqrt(app * aqq) || apq == 0.0) continue;
        rotated = true;
        const double zeta = (aqq - app) / (2.0 * apq);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < 3; ++i) {
          const double x = A[i][p], y = A[i][q];
          A[i][p] = c * x - sn * y; A[i][q] = sn * x + c * y;
          const double vx = W[i][p], vy = W[i][q];
          W[i][p] = c * vx - sn * vy; W[i][q] = sn * vx + c * vy;
        }
      }
    if (!rotated) break;
  }
  double nrm[3];
  int order[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) nrm[j] = std::sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) if (nrm[order[j]] > nrm[order[i]]) std::swap(order[i], order[j]);   // descending, stable
  const double tol = 1e-13 * std::max(nrm[order[0]], 1e-300);
  int have = 0;
  for (int c = 0; c < 3; ++c) {
    const int j = order[c];
    s[c] = nrm[j];
    for (int r = 0; r < 3; ++r) V.m[r][c] = W[r][j];
    if (nrm[j] > tol && have == c) {
      for (int r = 0; r < 3; ++r) U.m[r][c] = A[r][j] / nrm[j];
      have = c + 1;
    }
  }
  // complete U to an orthonormal basis where singular values vanish
  auto col = [&](int c, double out[3]) { for (int r = 0; r < 3; ++r) out[r] = U.m[r][c]; };
  auto set = [&](int c, const double in[3]) { for (int r = 0; r < 3; ++r) U.m[r][c] = in[r]; };
  if (have == 0) { U = Mat3d::identity(); return; }
  if (have == 1) {
    double u0[3]; col(0, u0);
    int k = std::fabs(u0[0]) < std::fabs(u0[1]) ? (std::fabs(u0[0]) < std::fabs(u0[2]) ? 0 : 2) : (std::fabs(u0[1]) < std::fabs(u0[2]) ? 1 : 2);
    double e[3] = {0, 0, 0}; e[k] = 1;
    const double d = u0[k];
    double u1[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
    const double n1 = std::sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    for (double& x : u1) x /= n1;
    set(1, u1);
    have = 2;
  }
  if (have == 2) {
    double u0[3], u1[3]; col(0, u0); col(1, u1);
    const double u2[3] = {u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2], u0[0] * u1[1] - u0[1] * u1[0]};
    set(2, u2);
  }
}

}  // namespace linalg
}  // namespace vo
