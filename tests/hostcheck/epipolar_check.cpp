// Host-only check of the facade's epipolar initialisation (include/vo/epipolar.hpp + vo/linalg.hpp): the
// eight-point fundamental, the essential matrix's two rotations and the cheirality vote, with the vote's
// triangulation count done by the same __host__ __device__ functors the GPU kernel runs (vo_math.h) -- so that
// `-m "not gpu"` covers this host logic of the product.  Built by tests/test_hostcheck.py with g++.
#include "../../visual-odometry_amd/csrc/vo_math.h"
#include "vo/epipolar.hpp"

extern "C" {

// K: 9 floats column-major; pairs: n x (i1, i2); p1 / p2: pixel coordinates; X_out: 16 floats column-major.
// Returns the number of correspondences in front of both cameras under the chosen candidate.
int hc_estimate_transform(const float* K, const int32_t* pairs, int n, const float* p1, int n1, const float* p2, int n2,
                          float* X_out) {
  vo::Matrix3f k;
  for (int i = 0; i < 9; ++i) k.m[i] = K[i];
  vo::IntPairVector corr((size_t)n);
  for (int i = 0; i < n; ++i) corr[(size_t)i] = vo::IntPair(pairs[2 * i], pairs[2 * i + 1]);
  vo::Vector2fVector a((size_t)n1), b((size_t)n2);
  for (int i = 0; i < n1; ++i) a[(size_t)i] = vo::Vector2f{{p1[2 * i], p1[2 * i + 1]}};
  for (int i = 0; i < n2; ++i) b[(size_t)i] = vo::Vector2f{{p2[2 * i], p2[2 * i + 1]}};
  int best = 0;
  const vo::Isometry3f X = vo::estimate_transform_with(k, corr, a, b, [&](const vo::Isometry3f& X_test) {
    // triangulate_points v1 (utils.cpp:51-76) on the host: the count of successful triangulations
    const vo::TriConst c = vo::tri_constants(k.data(), vo::pose_from_T16(X_test.data()));
    int cnt = 0;
    for (const auto& pr : corr) {
      const float h1[3] = {a[(size_t)pr.first][0], a[(size_t)pr.first][1], 1.f};
      const float h2[3] = {b[(size_t)pr.second][0], b[(size_t)pr.second][1], 1.f};
      float d1[3], d2[3], p[3];
      vo::mat3_vec(c.iK, 3, h1, d1);
      vo::mat3_vec(c.iRiK, 3, h2, d2);
      cnt += vo::triangulate_point(d1, d2, c.t, p) ? 1 : 0;
    }
    if (cnt > best) best = cnt;
    return cnt;
  });
  for (int i = 0; i < 16; ++i) X_out[i] = X.m[i];
  return best;
}

// jacobi_eigen_sym / svd3 of vo/linalg.hpp: M = U diag(s) V^T for a 3x3 (row-major in and out)
void hc_svd3(const double* M, double* U, double* s, double* V) {
  vo::linalg::Mat3d m, u, v;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.m[i][j] = M[3 * i + j];
  vo::linalg::svd3(m, u, s, v);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { U[3 * i + j] = u.m[i][j]; V[3 * i + j] = v.m[i][j]; }
}
}
