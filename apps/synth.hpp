// apps/synth.hpp -- seeded stand-ins for the reference's generators
// (utils.cpp:8-34), which draw from std::random_device and are therefore not
// reproducible.  Same distributions, splitmix64 stream.
#pragma once

#include <cmath>
#include <cstdint>

#include "vo/types.hpp"

namespace synth {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  float uniform(float lo, float hi) { return lo + (hi - lo) * (float)((next() >> 11) * (1.0 / 9007199254740992.0)); }
};

// generate_isometry3f (utils.cpp:8-20): axis ~ normalise(U(-1,1)^3), angle ~ U(-1,1), t ~ U(-1,1)^3
inline vo::Isometry3f generate_isometry3f(Rng& g, float max_angle = 1.f, float max_t = 1.f) {
  double a[3] = {g.uniform(-1, 1), g.uniform(-1, 1), g.uniform(-1, 1)};
  const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  for (double& x : a) x /= n;
  const double th = g.uniform(-max_angle, max_angle), c = std::cos(th), s = std::sin(th), C = 1 - c;
  vo::Isometry3f X = vo::Isometry3f::Identity();
  X(0, 0) = (float)(c + a[0] * a[0] * C);        X(0, 1) = (float)(a[0] * a[1] * C - a[2] * s); X(0, 2) = (float)(a[0] * a[2] * C + a[1] * s);
  X(1, 0) = (float)(a[1] * a[0] * C + a[2] * s); X(1, 1) = (float)(c + a[1] * a[1] * C);        X(1, 2) = (float)(a[1] * a[2] * C - a[0] * s);
  X(2, 0) = (float)(a[2] * a[0] * C - a[1] * s); X(2, 1) = (float)(a[2] * a[1] * C + a[0] * s); X(2, 2) = (float)(c + a[2] * a[2] * C);
  X(0, 3) = g.uniform(-max_t, max_t); X(1, 3) = g.uniform(-max_t, max_t); X(2, 3) = g.uniform(-max_t, max_t);
  return X;
}

// generate_points3d (utils.cpp:22-34): x,y ~ U(-10,10), z ~ U(-10,10)*0.1+1
inline vo::Vector3fVector generate_points3d(Rng& g, int n) {
  vo::Vector3fVector p((size_t)n);
  for (auto& q : p) { q[0] = g.uniform(-10, 10); q[1] = g.uniform(-10, 10); q[2] = g.uniform(-10, 10) * 0.1f + 1.0f; }
  return p;
}

}  // namespace synth
