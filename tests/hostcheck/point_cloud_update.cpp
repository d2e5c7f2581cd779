// PointCloudVector::update (PointCloud.h:52-66) keys on operator== of the appearances: -0 == +0, NaN equals nothing.
// Host-only check of the facade header (no GPU): compiled and run by tests/test_hostcheck.py.
#include <cstdio>
#include <cmath>
#include "vo/point_cloud.hpp"
using namespace vo;
int main() {
  PointCloudVector<3> map, c;
  c.resize(4);
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 10; ++k) c.appearances()[i].v[k] = (float)(i + k);
  c.appearances()[1].v[0] = 0.f; c.appearances()[2] = c.appearances()[1]; c.appearances()[2].v[0] = -0.f;   // -0 == +0
  c.appearances()[3].v[5] = NAN;
  for (int i = 0; i < 4; ++i) c.points()[i](0) = (float)i;
  map.update(c);            // entries: 0, 1 (overwritten by 2), 3(nan)
  map.update(c);            // 0 and 1 overwritten again; the NaN appearance is appended once more
  std::printf("%zu %g\n", map.size(), map.points()[1](0));
  return map.size() == 4 && map.points()[1](0) == 2.f ? 0 : 1;
}
