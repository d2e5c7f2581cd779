// vo/sequence.hpp -- the vo_complete frame loop (vo_complete.cpp:97-181) with every measurement set and every
// intermediate resident in HBM: an extension of the facade for callers that have the whole sequence (or a
// window of it) at hand.  All frames are uploaded once; the first pair is matched, initialised by
// vo_estimate_transform (the only host round trip) and triangulated; every later frame is
//     match -> join -> X_curr * model -> n x oneRound from the identity -> triangulate
// chained through device-side counts and the solver's device-side pose -- no host synchronisation per frame.
// The triangulated cloud of every frame stays on the device until cloud(t) fetches it (the map upkeep,
// PointCloudVector::update, is host code and runs afterwards).  Results are bit-identical to driving the same
// kernels frame by frame through Camera / PICPSolver / triangulate_points.
// setMatchUpFront(true): the matcher needs the appearances alone, so all F-1 consecutive pairs are matched by ONE
// vo_match_appearances_batch_dev call before the chain starts (frames of different sizes; the appearances are held as
// [F][capacity][10], so the pairs (t-1, t) are two views of one array); same pairs, same order, same results.
#pragma once

#include <algorithm>
#include <vector>

#include "camera.hpp"
#include "context.hpp"
#include "point_cloud.hpp"

namespace vo {

class DeviceSequence {
 public:
  DeviceSequence(const Camera& cam, const std::vector<PointCloudVector<2>>& frames, int rounds = 100,
                 float kernel_threshold = 10000.f)
      : ctx_(default_context().handle()), cam_(cam), F_((int)frames.size()), rounds_(rounds) {
    if (F_ < 2) throw Error(VO_ERR_INVALID_ARG, "DeviceSequence: need at least two measurement sets");
    off_.assign((size_t)F_ + 1, 0);
    for (int t = 0; t < F_; ++t) off_[(size_t)t + 1] = off_[(size_t)t] + frames[(size_t)t].size();
    cap_ = 1;
    for (const auto& f : frames) cap_ = std::max(cap_, f.size());
    std::vector<float> pts(2 * off_.back()), app(10 * cap_ * (size_t)F_, 0.f);
    for (int t = 0; t < F_; ++t) {
      const auto& f = frames[(size_t)t];
      if (f.size()) {
        std::memcpy(&pts[2 * off_[(size_t)t]], f.points()[0].data(), sizeof(float) * 2 * f.size());
        std::memcpy(&app[10 * cap_ * (size_t)t], f.appearances()[0].data(), sizeof(float) * 10 * f.size());
      }
    }
    first_[0] = frames[0].points(); first_[1] = frames[1].points();
    d_pts_ = upload(pts); d_app_ = upload(app);
    d_m_ = alloc<int32_t>(2 * cap_); d_j_ = alloc<int32_t>(2 * cap_); d_model_t_ = alloc<float>(3 * cap_);
    d_xyz_ = alloc<float>(3 * cap_ * (size_t)F_); d_pairs_ = alloc<int32_t>(2 * cap_ * (size_t)F_);
    d_tapp_ = alloc<float>(10 * cap_ * (size_t)F_);
    d_counts_ = alloc<int>(3 * (size_t)F_);
    d_traj_ = alloc<float>(16 * (size_t)F_);
    const std::vector<int> zeros(3 * (size_t)F_, 0);
    check(vo_memcpy_h2d(ctx_, d_counts_, zeros.data(), zeros.size() * sizeof(int)), "DeviceSequence");
    const Isometry3f I = Isometry3f::Identity();
    d_ident_ = upload(std::vector<float>(I.m, I.m + 16));
    check(vo_picp_create(ctx_, &solver_), "vo_picp_create");
    check(vo_picp_set_camera(solver_, cam.rows(), cam.cols(), cam.zNear(), cam.zFar(), cam.cameraMatrix().data(), I.data()),
          "vo_picp_set_camera");
    check(vo_picp_set_kernel_threshold(solver_, kernel_threshold), "vo_picp_set_kernel_threshold");
    check(vo_picp_pose_dev_ptr(solver_, &d_pose_), "vo_picp_pose_dev_ptr");
  }
  ~DeviceSequence() {
    if (solver_) vo_picp_destroy(solver_);
    for (void* d : owned_) vo_dev_free(ctx_, d);
  }
  DeviceSequence(const DeviceSequence&) = delete;
  DeviceSequence& operator=(const DeviceSequence&) = delete;

  int frames() const { return F_; }
  //! reference-order arithmetic for every solve of the chain (vo_picp_set_exact)
  void setExact(bool on) { check(vo_picp_set_exact(solver_, on ? 1 : 0), "vo_picp_set_exact"); }
  //! match every consecutive pair in one batched call at the start of run() instead of one call per frame inside the chain
  void setMatchUpFront(bool on) {
    if (on && !d_pm_) {
      std::vector<int> sizes((size_t)F_);
      for (int t = 0; t < F_; ++t) sizes[(size_t)t] = (int)n(t);
      d_n_all_ = alloc<int>((size_t)F_);
      check(vo_memcpy_h2d(ctx_, d_n_all_, sizes.data(), sizes.size() * sizeof(int)), "DeviceSequence::setMatchUpFront");
      d_pm_ = alloc<int32_t>(2 * cap_ * (size_t)(F_ - 1));
      d_pm_cnt_ = alloc<int>((size_t)(F_ - 1));
    }
    up_front_ = on;
  }

  //! enqueue the whole sequence; returns after the (host) epipolar initialisation, the chain runs on
  void run() {
    if (up_front_)
      check(vo_match_appearances_batch_dev(ctx_, F_ - 1, app_of(0), (int)cap_, d_n_all_, app_of(1), (int)cap_, d_n_all_ + 1, 0.1f,
                                           d_pm_, d_pm_cnt_), "vo_match_appearances_batch_dev");
    // first pair: vo_complete.cpp:121-132
    match(1);
    int c0 = 0;
    check(vo_memcpy_d2h(ctx_, &c0, cnt(1, 0), sizeof(int)), "DeviceSequence::run");
    std::vector<int32_t> pairs(2 * (size_t)std::max(c0, 1));
    if (c0) check(vo_memcpy_d2h(ctx_, pairs.data(), m_of(1), sizeof(int32_t) * 2 * (size_t)c0), "DeviceSequence::run");
    check(vo_estimate_transform(ctx_, cam_.cameraMatrix().data(), pairs.data(), c0, first_[0].empty() ? nullptr : first_[0][0].data(),
                                (int)first_[0].size(), first_[1].empty() ? nullptr : first_[1][0].data(), (int)first_[1].size(),
                                X0_.data()), "vo_estimate_transform");
    triangulate(1, X0_.data());
    const Isometry3f I = Isometry3f::Identity();
    check(vo_memcpy_h2d(ctx_, d_traj_, I.data(), 64), "DeviceSequence::run");
    check(vo_memcpy_h2d(ctx_, d_traj_ + 16, X0_.data(), 64), "DeviceSequence::run");
    // every later frame: vo_complete.cpp:150-179
    for (int t = 2; t < F_; ++t) {
      const int nq = (int)std::min(n(t - 1), n(t)), nq_prev = (int)std::min(n(t - 2), n(t - 1));
      match(t);
      check(vo_join_correspondences_dev(ctx_, m_of(t), nq, cnt(t, 0), pairs_of(t - 1), nq_prev, cnt(t - 1, 2), (int)n(t - 1),
                                        d_j_, cnt(t, 1)), "vo_join_correspondences_dev");
      check(vo_transform_points_dev(ctx_, t == 2 ? X0_.data() : nullptr, t == 2 ? nullptr : d_pose_, xyz_of(t - 1), nq_prev,
                                    cnt(t - 1, 2), d_model_t_), "vo_transform_points_dev");
      // the capacity (not the live count) sizes the solver's grid: one launch graph serves every frame
      check(vo_picp_set_points_dev(solver_, d_model_t_, (int)cap_, pts_of(t), (int)n(t)), "vo_picp_set_points_dev");
      check(vo_picp_set_pose_dev(solver_, d_ident_), "vo_picp_set_pose_dev");
      check(vo_picp_solve_dev(solver_, d_j_, (int)cap_, cnt(t, 1), 0, rounds_), "vo_picp_solve_dev");
      check(vo_picp_get_pose_dev(solver_, d_traj_ + 16 * (size_t)t), "vo_picp_get_pose_dev");
      triangulate(t, nullptr);
    }
  }

  //! poses of the trajectory (identity, X_1, X_2, ...); waits for the chain
  IsometryVector trajectory() const {
    IsometryVector out((size_t)F_);
    check(vo_memcpy_d2h(ctx_, out[0].data(), d_traj_, sizeof(float) * 16 * (size_t)F_), "DeviceSequence::trajectory");
    return out;
  }
  //! (matches, joined correspondences, triangulated points) of frame t
  void counts(int t, int& n_match, int& n_join, int& n_tri) const {
    int c[3];
    check(vo_memcpy_d2h(ctx_, c, d_counts_ + 3 * (size_t)t, sizeof(c)), "DeviceSequence::counts");
    if (up_front_ && t >= 1) check(vo_memcpy_d2h(ctx_, c, cnt(t, 0), sizeof(int)), "DeviceSequence::counts");
    n_match = c[0]; n_join = c[1]; n_tri = c[2];
  }
  //! triangulated cloud of frame t >= 1 (in the frame of camera t), with the appearances of frame t's points
  PointCloudVector<3> cloud(int t) const {
    int a, b, k;
    counts(t, a, b, k);
    PointCloudVector<3> pc((size_t)k);
    if (k) {
      check(vo_memcpy_d2h(ctx_, pc.points()[0].data(), xyz_of(t), sizeof(float) * 3 * (size_t)k), "DeviceSequence::cloud");
      check(vo_memcpy_d2h(ctx_, pc.appearances()[0].data(), d_tapp_ + 10 * cap_ * (size_t)t, sizeof(float) * 10 * (size_t)k),
            "DeviceSequence::cloud");
    }
    return pc;
  }
  int lastNumInliers() const { int k = 0; check(vo_picp_get_stats(solver_, nullptr, nullptr, &k), "vo_picp_get_stats"); return k; }

 private:
  template <class T>
  T* alloc(size_t count) {
    void* d = nullptr;
    check(vo_dev_alloc(ctx_, std::max<size_t>(count, 4) * sizeof(T), &d), "vo_dev_alloc");
    owned_.push_back(d);
    return static_cast<T*>(d);
  }
  float* upload(const std::vector<float>& h) {
    float* d = alloc<float>(h.size());
    if (!h.empty()) check(vo_memcpy_h2d(ctx_, d, h.data(), h.size() * sizeof(float)), "vo_memcpy_h2d");
    return d;
  }
  size_t n(int t) const { return off_[(size_t)t + 1] - off_[(size_t)t]; }
  const float* pts_of(int t) const { return d_pts_ + 2 * off_[(size_t)t]; }
  const float* app_of(int t) const { return d_app_ + 10 * cap_ * (size_t)t; }
  int* cnt(int t, int i) const { return (i == 0 && up_front_) ? d_pm_cnt_ + (t - 1) : d_counts_ + 3 * (size_t)t + i; }
  int32_t* m_of(int t) const { return up_front_ ? d_pm_ + 2 * cap_ * (size_t)(t - 1) : d_m_; }
  float* xyz_of(int t) const { return d_xyz_ + 3 * cap_ * (size_t)t; }
  int32_t* pairs_of(int t) const { return d_pairs_ + 2 * cap_ * (size_t)t; }
  void match(int t) {
    if (up_front_) return;                       // pairs and count of frame t are already where m_of / cnt point
    check(vo_match_appearances_dev(ctx_, app_of(t - 1), (int)n(t - 1), app_of(t), (int)n(t), 0.1f, d_m_, cnt(t, 0)),
          "vo_match_appearances_dev");
  }
  void triangulate(int t, const float* X_host) {
    const int nq = (int)std::min(n(t - 1), n(t));
    check(vo_triangulate_dev(ctx_, cam_.cameraMatrix().data(), X_host, X_host ? nullptr : d_pose_, m_of(t), nq, cnt(t, 0),
                             pts_of(t - 1), (int)n(t - 1), pts_of(t), (int)n(t), app_of(t), xyz_of(t), pairs_of(t),
                             d_tapp_ + 10 * cap_ * (size_t)t, cnt(t, 2)), "vo_triangulate_dev");
  }

  vo_ctx* ctx_;
  Camera cam_;
  int F_, rounds_;
  size_t cap_ = 1;
  std::vector<size_t> off_;
  Vector2fVector first_[2];
  Isometry3f X0_ = Isometry3f::Identity();
  vo_picp* solver_ = nullptr;
  const float* d_pose_ = nullptr;
  float *d_pts_ = nullptr, *d_app_ = nullptr, *d_model_t_ = nullptr, *d_xyz_ = nullptr, *d_tapp_ = nullptr, *d_traj_ = nullptr,
        *d_ident_ = nullptr;
  int32_t *d_m_ = nullptr, *d_j_ = nullptr, *d_pairs_ = nullptr, *d_pm_ = nullptr;
  int *d_counts_ = nullptr, *d_n_all_ = nullptr, *d_pm_cnt_ = nullptr;
  bool up_front_ = false;
  std::vector<void*> owned_;
};

}  // namespace vo
