// vo/types.hpp -- plain-old-data stand-ins for the Eigen types the reference
// passes across the Camera / PICPSolver boundary (defs.h:7-29).  Each one has
// exactly the memory of its Eigen counterpart (float, column-major, packed),
// so a std::vector of them *is* the array the C ABI (vo_hip.h) expects and a
// caller that does have Eigen can reinterpret its own vectors (vo/eigen_adaptor.hpp).
#pragma once

#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace vo {

template <int N>
struct Vecf {
  float v[N];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
  float& operator()(int i) { return v[i]; }
  const float& operator()(int i) const { return v[i]; }
  float& x() { return v[0]; }
  float& y() { return v[1]; }
  float& z() { static_assert(N >= 3, "no z"); return v[2]; }
  const float& x() const { return v[0]; }
  const float& y() const { return v[1]; }
  const float& z() const { static_assert(N >= 3, "no z"); return v[2]; }
  const float* data() const { return v; }
  float* data() { return v; }
  static Vecf Zero() { Vecf r; for (int i = 0; i < N; ++i) r.v[i] = 0.f; return r; }
  bool operator==(const Vecf& o) const { for (int i = 0; i < N; ++i) if (v[i] != o.v[i]) return false; return true; }
};
using Vector2f = Vecf<2>;
using Vector3f = Vecf<3>;
using Vector6f = Vecf<6>;
using Vector10f = Vecf<10>;

// column-major 3x3, the layout of Eigen::Matrix3f
struct Matrix3f {
  float m[9];
  float& operator()(int r, int c) { return m[r + 3 * c]; }
  const float& operator()(int r, int c) const { return m[r + 3 * c]; }
  const float* data() const { return m; }
  static Matrix3f Identity() { Matrix3f I{}; I(0, 0) = I(1, 1) = I(2, 2) = 1.f; return I; }
  // row-major initialiser, like Eigen's comma initialiser reads
  static Matrix3f FromRows(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
    Matrix3f M; M(0,0)=a; M(0,1)=b; M(0,2)=c; M(1,0)=d; M(1,1)=e; M(1,2)=f; M(2,0)=g; M(2,1)=h; M(2,2)=i; return M;
  }
};

// column-major 4x4 with last row 0 0 0 1, the layout of Eigen::Isometry3f
struct Isometry3f {
  float m[16];
  float& operator()(int r, int c) { return m[r + 4 * c]; }
  const float& operator()(int r, int c) const { return m[r + 4 * c]; }
  const float* data() const { return m; }
  float* data() { return m; }
  static Isometry3f Identity() { Isometry3f T{}; T(0,0)=T(1,1)=T(2,2)=T(3,3)=1.f; return T; }
  Vector3f translation() const { return Vector3f{{m[12], m[13], m[14]}}; }
  void setTranslation(const Vector3f& t) { m[12] = t[0]; m[13] = t[1]; m[14] = t[2]; }
  Matrix3f linear() const { Matrix3f R; for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R(r, c) = (*this)(r, c); return R; }
  void setLinear(const Matrix3f& R) { for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) (*this)(r, c) = R(r, c); }
  // host-side conveniences for drivers (cold path): R^T, -R^T t  and composition
  Isometry3f inverse() const {
    Isometry3f I = Identity();
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) I(r, c) = (*this)(c, r);
    for (int r = 0; r < 3; ++r) I(r, 3) = -(I(r, 0) * m[12] + (I(r, 1) * m[13] + I(r, 2) * m[14]));
    return I;
  }
  Isometry3f operator*(const Isometry3f& B) const {
    Isometry3f C = Identity();
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r)
      C(r, c) = (*this)(r, 0) * B(0, c) + ((*this)(r, 1) * B(1, c) + (*this)(r, 2) * B(2, c));
    for (int r = 0; r < 3; ++r)
      C(r, 3) = ((*this)(r, 0) * B(0, 3) + ((*this)(r, 1) * B(1, 3) + (*this)(r, 2) * B(2, 3))) + (*this)(r, 3);
    return C;
  }
};

using IntPair = std::pair<int, int>;                 // defs.h:18
using IntPairVector = std::vector<IntPair>;          // defs.h:19
using Vector2fVector = std::vector<Vector2f>;        // defs.h:27
using Vector3fVector = std::vector<Vector3f>;        // defs.h:25
using Vector10fVector = std::vector<Vector10f>;      // defs.h:26
using IsometryVector = std::vector<Isometry3f>;      // defs.h:29

static_assert(sizeof(Vector2f) == 8 && sizeof(Vector3f) == 12 && sizeof(Vector10f) == 40, "packed like Eigen");
static_assert(sizeof(Matrix3f) == 36 && sizeof(Isometry3f) == 64, "packed like Eigen");
static_assert(sizeof(IntPair) == 8, "IntPairVector must be int32 pairs for the C ABI");

inline const int32_t* pair_data(const IntPairVector& v) { return reinterpret_cast<const int32_t*>(v.data()); }
inline int32_t* pair_data(IntPairVector& v) { return reinterpret_cast<int32_t*>(v.data()); }

}  // namespace vo
