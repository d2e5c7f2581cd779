// vo/kdtree.hpp -- the reference's TreeNode_ (include/eigen_kdtree.h:6-121) over libvo_hip.so: one object per point set,
// the four query modes for whole query sets.  The reference hands back pointers into the caller's (reordered) point
// vector; here every answer is the INDEX of the point in the vector the tree was built from.
//   bestMatchFull / fullSearch : exact, tree-independent (vo_match-style scan / vo_radius_search)
//   bestMatchFast / fastSearch : one side of every PCA split, then the leaf (vo_kdtree_*): answers depend on the tree,
//                                which is built like the reference's (eigen_kdtree.h:18-38)
#pragma once

#include <vector>

#include "context.hpp"
#include "types.hpp"
#include "utils.hpp"

namespace vo {

class KdTree {
 public:
  //! TreeNode_(begin, end, max_points_in_leaf = 20), eigen_kdtree.h:18-38
  explicit KdTree(const Vector10fVector& points, int max_points_in_leaf = 20) : points_(points) {
    check(vo_kdtree_create(default_context().handle(), points.empty() ? nullptr : points[0].data(), (int)points.size(),
                           max_points_in_leaf, &h_), "KdTree");
  }
  ~KdTree() { vo_kdtree_destroy(h_); }
  KdTree(const KdTree&) = delete;
  KdTree& operator=(const KdTree&) = delete;

  //! eigen_kdtree.h:75-85 for every query: index of the closest point of the query's leaf within `norm`, or -1
  std::vector<int> bestMatchFast(const Vector10fVector& queries, float norm) const {
    std::vector<int> out(queries.size(), -1);
    check(vo_kdtree_best_match_fast(h_, queries.empty() ? nullptr : queries[0].data(), (int)queries.size(), norm,
                                    out.empty() ? nullptr : out.data()), "KdTree::bestMatchFast");
    return out;
  }
  //! eigen_kdtree.h:40-52 for every query: the points of the query's leaf within `norm`, in leaf order
  std::vector<std::vector<int>> fastSearch(const Vector10fVector& queries, float norm) const {
    const int nq = (int)queries.size();
    std::vector<int32_t> off((size_t)nq + 1, 0), idx((size_t)std::max(2 * nq, 16));
    for (;;) {
      int total = 0;
      const int rc = vo_kdtree_fast_search(h_, nq ? queries[0].data() : nullptr, nq, norm, off.data(), idx.data(), (int)idx.size(), &total);
      if (rc == VO_ERR_INVALID_ARG && total > (int)idx.size()) { idx.resize((size_t)total); continue; }
      check(rc, "KdTree::fastSearch");
      break;
    }
    std::vector<std::vector<int>> out((size_t)nq);
    for (int i = 0; i < nq; ++i) out[(size_t)i].assign(idx.begin() + off[(size_t)i], idx.begin() + off[(size_t)i + 1]);
    return out;
  }
  //! eigen_kdtree.h:90-115 for every query: exact nearest point within `norm` (index or -1)
  std::vector<int> bestMatchFull(const Vector10fVector& queries, float norm) const {
    // vo_match_appearances searches the LARGER set; here the roles are fixed, so go through the radius search and keep
    // the closest hit (lowest index on exact ties, as the library's matcher does)
    const std::vector<std::vector<int>> hits = full_search(points_, queries, norm);
    std::vector<int> out(queries.size(), -1);
    for (size_t i = 0; i < queries.size(); ++i) {
      float best = norm * norm;
      for (int j : hits[i]) {
        float s = 0.f;
        for (int k = 0; k < 10; ++k) { const float d = points_[(size_t)j](k) - queries[i](k); s += d * d; }
        if (s < best || (s == best && out[i] >= 0 && j < out[i])) { best = s; out[i] = j; }
      }
    }
    return out;
  }
  //! eigen_kdtree.h:56-71 for every query: every point within `norm`
  std::vector<std::vector<int>> fullSearch(const Vector10fVector& queries, float norm) const { return full_search(points_, queries, norm); }

  //! (points, nodes, leaves) of the tree
  void info(int& n_points, int& n_nodes, int& n_leaves) const { check(vo_kdtree_info(h_, &n_points, &n_nodes, &n_leaves), "KdTree::info"); }
  int size() const { int n, a, b; info(n, a, b); return n; }
  int nodes() const { int n, a, b; info(n, a, b); return a; }
  int leaves() const { int n, a, b; info(n, a, b); return b; }

  vo_kdtree* handle() const { return h_; }

 private:
  vo_kdtree* h_ = nullptr;
  Vector10fVector points_;
};

}  // namespace vo
