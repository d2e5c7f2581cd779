// vo/point_cloud.hpp -- PointCloud<dim> / PointCloudVector<dim> (PointCloud.h:9-75):
// structure-of-arrays container, points + 10-D appearances.  The const
// accessors return references (the reference returns whole-vector copies,
// PointCloud.h:70-71, which makes its triangulation O(N^2) in bytes).
#pragma once

#include <cstring>
#include <string>
#include <unordered_map>

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
  //! map upsert keyed by exact appearance equality (PointCloud.h:52-66): a point whose
  //! appearance is already present overwrites that entry (the FIRST such entry, as the
  //! reference's inner loop stops at the first hit), otherwise it is appended.  The
  //! reference scans linearly (O(N*M)); the same result is obtained with a hash index
  //! of first occurrences, rebuilt when the container was modified behind its back.
  void update(const PointCloudVector<dim>& cloud) {
    if (_index_size != _appearances.size()) {
      _index.clear();
      for (size_t j = 0; j < _appearances.size(); ++j)
        if (!has_nan(_appearances[j])) _index.emplace(key(_appearances[j]), j);
      _index_size = _appearances.size();
    }
    for (size_t i = 0; i < cloud.size(); i++) {
      const Vector10f& a = cloud.appearances()[i];
      // the reference compares with operator== (PointCloud.h:56): an appearance with a NaN equals nothing, not even itself
      const bool nan = has_nan(a);
      auto it = nan ? _index.end() : _index.find(key(a));
      if (it != _index.end()) {
        _points[it->second] = cloud.points()[i];
      } else {
        if (!nan) _index.emplace(key(a), _points.size());
        _points.push_back(cloud.points()[i]);
        _appearances.push_back(a);
        _index_size = _appearances.size();
      }
    }
  }
  PointsVec& points() { return _points; }
  Vector10fVector& appearances() { return _appearances; }
  const PointsVec& points() const { return _points; }
  const Vector10fVector& appearances() const { return _appearances; }

 protected:
  static bool has_nan(const Vector10f& a) {
    for (int k = 0; k < 10; ++k) if (a.v[k] != a.v[k]) return true;
    return false;
  }
  static std::string key(const Vector10f& a) {
    // the equivalence of operator== on floats: -0 == +0 (both keyed as +0), NaN != NaN (never keyed: has_nan)
    float c[10];
    for (int k = 0; k < 10; ++k) c[k] = a.v[k] == 0.f ? 0.f : a.v[k];
    return std::string(reinterpret_cast<const char*>(c), sizeof(c));
  }
  PointsVec _points;
  Vector10fVector _appearances;
  std::unordered_map<std::string, size_t> _index;
  size_t _index_size = 0;
};

}  // namespace vo
