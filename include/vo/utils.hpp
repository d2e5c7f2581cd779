// vo/utils.hpp -- free functions of the hot path with the reference's
// signatures: triangulate_points x3 (utils.h:131-160), the image matcher and
// the index join (vo_complete.cpp:12-66), Isometry * point cloud
// (PointCloud.h:77-82).
#pragma once

#include "context.hpp"
#include "point_cloud.hpp"
#include "types.hpp"

namespace vo {

namespace detail {
inline int triangulate(const Matrix3f& k, const Isometry3f& X, const IntPairVector& corr, const float* p1, int n1,
                       const float* p2, int n2, const float* app2, Vector3fVector& tri, IntPairVector* corr_new,
                       Vector10fVector* app_out) {
  const int n = static_cast<int>(corr.size());
  tri.resize(corr.size());
  if (corr_new) corr_new->resize(corr.size());
  if (app_out) app_out->resize(corr.size());
  int n_out = 0;
  check(vo_triangulate(default_context().handle(), k.data(), X.data(), pair_data(corr), n, p1, n1, p2, n2,
                       app_out ? app2 : nullptr, n ? tri[0].data() : nullptr,
                       corr_new && n ? pair_data(*corr_new) : nullptr,
                       app_out && n ? (*app_out)[0].data() : nullptr, &n_out), "triangulate_points");
  tri.resize(static_cast<size_t>(n_out));
  if (corr_new) corr_new->resize(static_cast<size_t>(n_out));
  if (app_out) app_out->resize(static_cast<size_t>(n_out));
  return n_out;
}
inline const float* ptr(const Vector2fVector& v) { return v.empty() ? nullptr : v[0].data(); }
}  // namespace detail

//! utils.cpp:51-76 -- points only
inline int triangulate_points(const Matrix3f& k, const Isometry3f& X, const IntPairVector& correspondences,
                              const Vector2fVector& p1_img, const Vector2fVector& p2_img, Vector3fVector& triangulated) {
  return detail::triangulate(k, X, correspondences, detail::ptr(p1_img), (int)p1_img.size(), detail::ptr(p2_img),
                             (int)p2_img.size(), nullptr, triangulated, nullptr, nullptr);
}
//! utils.cpp:77-105 -- plus (index in second image, index of triangulated point)
inline int triangulate_points(const Matrix3f& k, const Isometry3f& X, const IntPairVector& correspondences,
                              const Vector2fVector& p1_img, const Vector2fVector& p2_img, Vector3fVector& triangulated,
                              IntPairVector& correspondences_new) {
  return detail::triangulate(k, X, correspondences, detail::ptr(p1_img), (int)p1_img.size(), detail::ptr(p2_img),
                             (int)p2_img.size(), nullptr, triangulated, &correspondences_new, nullptr);
}
//! utils.cpp:106-134 -- point clouds: the appearance of the second image's point rides along
inline int triangulate_points(const Matrix3f& k, const Isometry3f& X, const IntPairVector& correspondences,
                              const PointCloudVector<2>& pc_1, const PointCloudVector<2>& pc_2,
                              PointCloudVector<3>& triangulated, IntPairVector& correspondences_new) {
  const Vector10fVector& a2 = pc_2.appearances();
  return detail::triangulate(k, X, correspondences, detail::ptr(pc_1.points()), (int)pc_1.size(),
                             detail::ptr(pc_2.points()), (int)pc_2.size(), a2.empty() ? nullptr : a2[0].data(),
                             triangulated.points(), &correspondences_new, &triangulated.appearances());
}

//! vo_complete.cpp:12-49 -- pairs (ref_idx, curr_idx); exact NN within 0.1 in appearance space
inline IntPairVector compute_correspondences_images(const Vector10fVector& appearances1,
                                                    const Vector10fVector& appearances2) {
  const int n1 = (int)appearances1.size(), n2 = (int)appearances2.size();
  IntPairVector out(static_cast<size_t>(n1 < n2 ? n1 : n2));
  int n_out = 0;
  check(vo_match_appearances(default_context().handle(), n1 ? appearances1[0].data() : nullptr, n1,
                             n2 ? appearances2[0].data() : nullptr, n2, 0.1f, out.empty() ? nullptr : pair_data(out),
                             &n_out), "compute_correspondences_images");
  out.resize(static_cast<size_t>(n_out));
  return out;
}

//! TreeNode_::fullSearch (eigen_kdtree.h:56-71) for every query at once: answers[i] = indices of ALL points of
//! `set` closer than `norm` to queries[i] (strict <, as bruteForceSearch); the order inside a list is unspecified
inline std::vector<std::vector<int>> full_search(const Vector10fVector& set, const Vector10fVector& queries, float norm) {
  const int nt = (int)set.size(), nq = (int)queries.size();
  std::vector<int32_t> offsets((size_t)nq + 1, 0), indices((size_t)(nq > 8 ? 2 * nq : 16));
  int n_total = 0;
  for (;;) {
    const int rc = vo_radius_search(default_context().handle(), nt ? set[0].data() : nullptr, nt,
                                    nq ? queries[0].data() : nullptr, nq, norm, offsets.data(), indices.data(),
                                    (int)indices.size(), &n_total);
    if (rc == VO_ERR_INVALID_ARG && n_total > (int)indices.size()) { indices.resize((size_t)n_total); continue; }
    check(rc, "full_search");
    break;
  }
  std::vector<std::vector<int>> answers((size_t)nq);
  for (int i = 0; i < nq; ++i) answers[(size_t)i].assign(indices.begin() + offsets[(size_t)i], indices.begin() + offsets[(size_t)i + 1]);
  return answers;
}

//! vo_complete.cpp:52-66 -- (ref,curr) join (ref,world) -> (curr,world), first partner wins
inline IntPairVector extract_correspondences_world(const IntPairVector& correspondences_imgs,
                                                   const IntPairVector& correspondences_world) {
  IntPairVector out(correspondences_imgs.size());
  int n_out = 0;
  check(vo_join_correspondences(default_context().handle(), pair_data(correspondences_imgs),
                                (int)correspondences_imgs.size(), pair_data(correspondences_world),
                                (int)correspondences_world.size(), out.empty() ? nullptr : pair_data(out), &n_out),
        "extract_correspondences_world");
  out.resize(static_cast<size_t>(n_out));
  return out;
}

//! X * points (vo_daKnown.cpp:144-145)
inline Vector3fVector transform_points(const Isometry3f& X, const Vector3fVector& pts) {
  Vector3fVector out(pts.size());
  check(vo_transform_points(default_context().handle(), X.data(), pts.empty() ? nullptr : pts[0].data(),
                            (int)pts.size(), out.empty() ? nullptr : out[0].data()), "transform_points");
  return out;
}

//! PointCloud.h:77-82 -- appearances are carried through unchanged
inline PointCloudVector<3> operator*(const Isometry3f& X, const PointCloudVector<3>& pc) {
  PointCloudVector<3> ret;
  ret.points() = transform_points(X, pc.points());
  ret.appearances() = pc.appearances();
  return ret;
}

}  // namespace vo
