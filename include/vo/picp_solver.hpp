// vo/picp_solver.hpp -- the reference's PICPSolver (picp_solver.h:18-79) over
// libvo_hip.so: same construction, init, kernelThreshold accessors, oneRound,
// camera(), chiInliers(), chiOutliers(), numInliers().
//
// Differences a caller can observe:
//  * oneRound() enqueues the Gauss-Newton round on the GPU and returns at once;
//    camera()/chi*()/numInliers() wait for it.  It returns true always: the
//    reference can only return false when inliers < _min_num_inliers, which is
//    0 and has no setter (picp_solver.cpp:11,103-107).
//  * init() copies the point vectors to the GPU (the reference keeps raw
//    pointers, picp_solver.cpp:21-22): changing them afterwards needs a new init().
//  * every oneRound honours the vector it is given, like the reference
//    (picp_solver.cpp:62): the library compares it in full with its GPU copy and
//    uploads it again when anything changed (in-place edits included);
//    solve(corr, keep, n) runs n rounds with no host round trip at all.
//  * setExact(true) switches the handle to reference-order arithmetic: results
//    bit-identical to the reference's scalar float32 loop (vo_picp_set_exact).
#pragma once

#include "camera.hpp"

namespace vo {

class PICPSolver {
 public:
  PICPSolver() { check(vo_picp_create(default_context().handle(), &h_), "vo_picp_create"); }
  ~PICPSolver() { vo_picp_destroy(h_); }
  // copyable like the reference's (picp_solver.h has no copy restrictions): a copy is a fresh device twin with the
  // same settings; like the reference's raw point-vector pointers, the points need a new init() to be shared safely
  PICPSolver(const PICPSolver& o) : _camera(o._camera), exact_(o.exact_) {
    check(vo_picp_create(default_context().handle(), &h_), "vo_picp_create");
    setKernelThreshold(o.kernelThreshold());
    setExact(o.exact_);
  }
  PICPSolver& operator=(const PICPSolver& o) {
    if (this != &o) { _camera = o._camera; setKernelThreshold(o.kernelThreshold()); setExact(o.exact_); }
    return *this;
  }

  //! init method, call it at the beginning (picp_solver.cpp:16-23)
  void init(const Camera& camera, const Vector3fVector& world_points, const Vector2fVector& image_points) {
    _camera = camera;
    check(vo_picp_set_camera(h_, camera.rows(), camera.cols(), camera.zNear(), camera.zFar(),
                             camera.cameraMatrix().data(), camera.worldInCameraPose().data()), "PICPSolver::init");
    check(vo_picp_set_points(h_, world_points.empty() ? nullptr : world_points[0].data(),
                             static_cast<int>(world_points.size()),
                             image_points.empty() ? nullptr : image_points[0].data(),
                             static_cast<int>(image_points.size())), "PICPSolver::init");
  }

  float kernelThreshold() const { float t = 0; check(vo_picp_get_kernel_threshold(h_, &t), "kernelThreshold"); return t; }
  void setKernelThreshold(float kernel_threshold) { check(vo_picp_set_kernel_threshold(h_, kernel_threshold), "setKernelThreshold"); }

  //! accessor to the camera: synchronises and refreshes the pose
  const Camera& camera() const {
    Isometry3f T;
    check(vo_picp_get_pose(h_, T.data()), "PICPSolver::camera");
    _camera.setWorldInCameraPose(T);
    return _camera;
  }
  float chiInliers() const { float a = 0; check(vo_picp_get_stats(h_, &a, nullptr, nullptr), "chiInliers"); return a; }
  float chiOutliers() const { float a = 0; check(vo_picp_get_stats(h_, nullptr, &a, nullptr), "chiOutliers"); return a; }
  int numInliers() const { int n = 0; check(vo_picp_get_stats(h_, nullptr, nullptr, &n), "numInliers"); return n; }

  //! one Gauss-Newton iteration (picp_solver.cpp:98-112); pairs are (measurement, model)
  bool oneRound(const IntPairVector& correspondences, bool keep_outliers) {
    check(vo_picp_one_round(h_, pair_data(correspondences), static_cast<int>(correspondences.size()),
                            keep_outliers ? 1 : 0), "PICPSolver::oneRound");
    return true;
  }
  //! n_iters rounds back to back on the GPU (extension)
  void solve(const IntPairVector& correspondences, bool keep_outliers, int n_iters) {
    check(vo_picp_solve(h_, pair_data(correspondences), static_cast<int>(correspondences.size()),
                        keep_outliers ? 1 : 0, n_iters), "PICPSolver::solve");
  }
  //! reference-order arithmetic: bit-identical to the reference's float32 loop (extension, off by default)
  void setExact(bool on) { check(vo_picp_set_exact(h_, on ? 1 : 0), "PICPSolver::setExact"); exact_ = on; }
  bool exact() const { return exact_; }
  vo_picp* handle() const { return h_; }

 protected:
  vo_picp* h_ = nullptr;
  mutable Camera _camera;
  bool exact_ = false;
};

}  // namespace vo
