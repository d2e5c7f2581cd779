// vo/point_cloud.hpp -- PointCloud<dim> / PointCloudVector<dim> (PointCloud.h:9-75):
// structure-of-arrays container, points + 10-D appearances.  The const
// accessors return references (the reference returns whole-vector copies,
// PointCloud.h:70-71, which makes its triangulation O(N^2) in bytes).
#pragma once

#include "types.hpp"

namespace vo {

template <int dim>
class PointCloud {
 public:
  PointCloud() {}
  PointCloud(const Vecf<dim>& point, const Vector10f& appearance) : _point(point), _appearance(appearance) {}
  const Vecf<dim>& point() const { return _point; }
  const Vector10f& appearance() const { return _appearance; }

 protected:
  Vecf<dim> _point;
  Vector10f _appearance;
};

template <int dim>
class PointCloudVector {
 public:
  using PointsVec = std::vector<Vecf<dim>>;
  PointCloudVector() {}
  explicit PointCloudVector(size_t N) : _points(N), _appearances(N) {}
  void push_back(const PointCloud<dim>& pc) { _points.push_back(pc.point()); _appearances.push_back(pc.appearance()); }
  void resize(size_t N) { _points.resize(N); _appearances.resize(N); }
  void reserve(size_t N) { _points.reserve(N); _appearances.reserve(N); }
  size_t size() const { return _points.size(); }
  void clear() { _points.clear(); _appearances.clear(); }
  PointsVec& points() { return _points; }
  Vector10fVector& appearances() { return _appearances; }
  const PointsVec& points() const { return _points; }
  const Vector10fVector& appearances() const { return _appearances; }

 protected:
  PointsVec _points;
  Vector10fVector _appearances;
};

}  // namespace vo
