// vo/sequence.hpp -- the vo_complete frame loop (vo_complete.cpp:97-181) with every measurement set and every
// intermediate resident in HBM: an extension of the facade for callers that have the whole sequence (or a
// window of it) at hand.  All frames are uploaded once; the first pair is matched, initialised by
// vo_estimate_transform (the only host round trip) and triangulated; every later frame is
//     match -> join -> X_curr * model -> n x oneRound from the identity -> triangulate
// chained through device-side counts and the solver's device-side pose -- no host synchronisation per frame.
// The triangulated cloud of every frame stays on the device until cloud(t) fetches it.  setKeepMap(true): the map upkeep
// of the loop body -- map.update(history * triangulated_pc), history = history * pose^-1 (vo_complete.cpp:145-147,
// 175-176; PointCloud.h:52-66) -- runs on the device too, inside the chain (vo_map_*: a hash table of first
// occurrences instead of the reference's O(N M) scan; same entries in the same order); map() fetches it.  Results are
// bit-identical to driving the same kernels frame by frame through Camera / PICPSolver / triangulate_points.
// setMatchUpFront(true): the matcher needs the appearances alone, so all F-1 consecutive pairs are matched by ONE
// vo_match_appearances_batch_dev call before the chain starts (frames of different sizes; the appearances are held as
// [F][capacity][10], so the pairs (t-1, t) are two views of one array; more than 65535 pairs go in several calls); same
// pairs, same order, same results.  setMatchesExternal(): the pairs come from somewhere else altogether -- the native
// multi-GPU driver matches blocks of consecutive pairs on the node's GPUs and gathers them (apps/sequence_mgpu.cpp,
// SURVEY 8(e)) -- as [row][capacity] pairs + one count per row in device memory of this context, row_of[t - 1] naming
// the row of pair (t-1, t).  The mode in force when run() starts is the one counts() / cloud() read afterwards.
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
    if (map_) vo_map_destroy(map_);
    if (solver_) vo_picp_destroy(solver_);
    for (void* d : owned_) vo_dev_free(ctx_, d);
  }
  DeviceSequence(const DeviceSequence&) = delete;
  DeviceSequence& operator=(const DeviceSequence&) = delete;

  int frames() const { return F_; }
  //! reference-order arithmetic for every solve of the chain (vo_picp_set_exact)
  void setExact(bool on) { check(vo_picp_set_exact(solver_, on ? 1 : 0), "vo_picp_set_exact"); }
  //! keep the map inside the chain (vo_complete.cpp:145-147,175-176), on the device; capacity: entries to make room for up
  //! front (default: every point of the sequence could be a new landmark, up to 8 frames' worth -- the map grows beyond)
  void setKeepMap(bool on, size_t capacity = 0) {
    if (ran_) throw Error(VO_ERR_INVALID_ARG, "DeviceSequence: setKeepMap after run()");
    if (on && !map_) {
      const size_t cap = capacity ? capacity : std::min(off_.back(), 8 * cap_);
      check(vo_map_create(ctx_, (int)std::min<size_t>(cap, (size_t)1 << 29), &map_), "vo_map_create");
    }
    keep_map_ = on;
  }
  //! match every consecutive pair in one batched call at the start of run() instead of one call per frame inside the chain
  void setMatchUpFront(bool on) {
    if (ran_) throw Error(VO_ERR_INVALID_ARG, "DeviceSequence: the matching mode cannot change after run()");
    if (on && !d_pm_own_) {
      std::vector<int> sizes((size_t)F_);
      for (int t = 0; t < F_; ++t) sizes[(size_t)t] = (int)n(t);
      d_n_all_ = alloc<int>((size_t)F_);
      check(vo_memcpy_h2d(ctx_, d_n_all_, sizes.data(), sizes.size() * sizeof(int)), "DeviceSequence::setMatchUpFront");
      d_pm_own_ = alloc<int32_t>(2 * cap_ * (size_t)(F_ - 1));
      d_pm_cnt_own_ = alloc<int>((size_t)(F_ - 1));
    }
    mode_ = on ? UpFront : PerFrame;
  }
  //! the pairs of every consecutive frame pair, computed elsewhere: d_pairs[row][capacity()] pairs (ref_idx, cur_idx) and
  //! d_counts[row] in device memory of this context; pair (t-1, t) sits in row row_of[t - 1]  (row_of.size() == frames() - 1)
  void setMatchesExternal(const int32_t* d_pairs, const int* d_counts, const std::vector<int>& row_of) {
    if (ran_) throw Error(VO_ERR_INVALID_ARG, "DeviceSequence: the matching mode cannot change after run()");
    if ((int)row_of.size() != F_ - 1 || !d_pairs || !d_counts) throw Error(VO_ERR_INVALID_ARG, "DeviceSequence::setMatchesExternal: one row per consecutive pair");
    d_pm_ext_ = d_pairs; d_pm_cnt_ext_ = d_counts; row_of_ = row_of;
    mode_ = External;
  }
  //! pairs per row of the arrays setMatchesExternal takes (the largest measurement set)
  size_t capacity() const { return cap_; }

  //! enqueue the whole sequence; returns after the (host) epipolar initialisation, the chain runs on
  void run() {
    ran_ = true;
    if (mode_ == UpFront) {
      // the frame is a grid dimension of the batched matcher: at most 65535 pairs per call
      for (int p0 = 0; p0 < F_ - 1; p0 += 65535) {
        const int np = std::min(65535, F_ - 1 - p0);
        check(vo_match_appearances_batch_dev(ctx_, np, app_of(p0), (int)cap_, d_n_all_ + p0, app_of(p0 + 1), (int)cap_, d_n_all_ + p0 + 1,
                                             0.1f, d_pm_own_ + 2 * cap_ * (size_t)p0, d_pm_cnt_own_ + p0), "vo_match_appearances_batch_dev");
      }
    }
    // first pair: vo_complete.cpp:121-132
    match(1);
    check(vo_estimate_transform_dev(ctx_, cam_.cameraMatrix().data(), m_of(1), (int)std::min(n(0), n(1)), cnt(1, 0), pts_of(0), (int)n(0),
                                    pts_of(1), (int)n(1), X0_.data()), "vo_estimate_transform_dev");
    triangulate(1, X0_.data());
    const Isometry3f I = Isometry3f::Identity();
    check(vo_memcpy_h2d(ctx_, d_traj_, I.data(), 64), "DeviceSequence::run");
    check(vo_memcpy_h2d(ctx_, d_traj_ + 16, X0_.data(), 64), "DeviceSequence::run");
    if (keep_map_) {                                             // vo_complete.cpp:145-146
      check(vo_map_clear(map_), "vo_map_clear");
      map_update(1, nullptr);
      check(vo_map_history_reset_dev(map_, d_traj_ + 16), "vo_map_history_reset_dev");
      check(vo_map_history_dev_ptr(map_, &d_history_), "vo_map_history_dev_ptr");
    }
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
      if (keep_map_) {                                           // vo_complete.cpp:175-176
        map_update(t, d_history_);
        check(vo_map_history_step_dev(map_, d_pose_), "vo_map_history_step_dev");
      }
    }
  }

  //! the map (setKeepMap): entries in the order the reference's update leaves them; H (or null): map = H * map first
  //! (vo_complete.cpp:183; applied once, in place).  Waits for the chain.
  PointCloudVector<3> map(const Isometry3f* H = nullptr) {
    if (!keep_map_ || !map_) throw Error(VO_ERR_NOT_READY, "DeviceSequence::map: setKeepMap(true) before run()");
    if (H) check(vo_map_transform(map_, H->data()), "vo_map_transform");
    int k = 0;
    check(vo_map_size(map_, &k), "vo_map_size");
    PointCloudVector<3> pc((size_t)k);
    if (k) check(vo_map_read(map_, pc.points()[0].data(), pc.appearances()[0].data(), k, &k), "vo_map_read");
    return pc;
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
    if (mode_ != PerFrame && t >= 1) check(vo_memcpy_d2h(ctx_, c, cnt(t, 0), sizeof(int)), "DeviceSequence::counts");
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
  // where the pairs of (t-1, t) and their count live: the chain's own buffers, the up-front call's, or the caller's rows
  int* cnt(int t, int i) const {
    if (i == 0 && mode_ == UpFront) return d_pm_cnt_own_ + (t - 1);
    if (i == 0 && mode_ == External) return const_cast<int*>(d_pm_cnt_ext_) + row_of_[(size_t)t - 1];
    return d_counts_ + 3 * (size_t)t + i;
  }
  int32_t* m_of(int t) const {
    if (mode_ == UpFront) return d_pm_own_ + 2 * cap_ * (size_t)(t - 1);
    if (mode_ == External) return const_cast<int32_t*>(d_pm_ext_) + 2 * cap_ * (size_t)row_of_[(size_t)t - 1];
    return d_m_;
  }
  float* xyz_of(int t) const { return d_xyz_ + 3 * cap_ * (size_t)t; }
  int32_t* pairs_of(int t) const { return d_pairs_ + 2 * cap_ * (size_t)t; }
  void match(int t) {
    if (mode_ != PerFrame) return;               // pairs and count of frame t are already where m_of / cnt point
    check(vo_match_appearances_dev(ctx_, app_of(t - 1), (int)n(t - 1), app_of(t), (int)n(t), 0.1f, d_m_, cnt(t, 0)),
          "vo_match_appearances_dev");
  }
  void map_update(int t, const float* d_T16) {
    const int nq = (int)std::min(n(t - 1), n(t));
    check(vo_map_update_dev(map_, xyz_of(t), d_tapp_ + 10 * cap_ * (size_t)t, nq, cnt(t, 2), d_T16), "vo_map_update_dev");
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
  Isometry3f X0_ = Isometry3f::Identity();
  vo_picp* solver_ = nullptr;
  vo_map* map_ = nullptr;
  bool keep_map_ = false;
  const float* d_history_ = nullptr;
  const float* d_pose_ = nullptr;
  float *d_pts_ = nullptr, *d_app_ = nullptr, *d_model_t_ = nullptr, *d_xyz_ = nullptr, *d_tapp_ = nullptr, *d_traj_ = nullptr,
        *d_ident_ = nullptr;
  int32_t *d_m_ = nullptr, *d_j_ = nullptr, *d_pairs_ = nullptr, *d_pm_own_ = nullptr;
  int *d_counts_ = nullptr, *d_n_all_ = nullptr, *d_pm_cnt_own_ = nullptr;
  const int32_t* d_pm_ext_ = nullptr;
  const int* d_pm_cnt_ext_ = nullptr;
  std::vector<int> row_of_;
  enum Mode { PerFrame, UpFront, External } mode_ = PerFrame;
  bool ran_ = false;
  std::vector<void*> owned_;
};

}  // namespace vo
