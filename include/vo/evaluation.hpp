// vo/evaluation.hpp -- the reference's offline metrics (evaluate.cpp:7-90,
// evaluation_utils.cpp): orientation error trace(I - R_rel^T R_rel,gt) and
// translation ratio of consecutive relative poses, the inverse median ratio as
// the monocular scale, RMSE of the scaled trajectory and of the scaled map
// against the landmarks with the same appearance.
#pragma once

#include <algorithm>
#include <cmath>
#include <fstream>
#include <sstream>
#include <unordered_map>

#include "files.hpp"

namespace vo {

template <int dim>
inline std::vector<Vecf<dim>> read_eigen_vectors(const std::string& file_path) {   // evaluation_utils.h:6-30
  std::vector<Vecf<dim>> pts;
  std::ifstream in(file_path);
  if (!in.is_open()) { std::cout << "Unable to open " << file_path << std::endl; return pts; }
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    Vecf<dim> p;
    for (int i = 0; i < dim; i++) ss >> p[i];
    pts.push_back(p);
  }
  return pts;
}

inline float median(std::vector<float> v) {                                          // evaluation_utils.cpp:65-70
  const size_t n = v.size() / 2;
  std::nth_element(v.begin(), v.begin() + (long)n, v.end());
  return v[n];
}

inline IsometryVector get_gt_data(const std::string& file_path) {                   // evaluation_utils.cpp:3-31
  IsometryVector data;
  std::ifstream in(file_path);
  if (!in.is_open()) { std::cout << "Unable to open " << file_path << std::endl; return data; }
  std::string line, word;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    for (int i = 0; i < 4; i++) ss >> word;
    float x = 0, y = 0, th = 0;
    ss >> x >> y >> th;
    Isometry3f X = Isometry3f::Identity();
    const float s = std::sin(th), c = std::cos(th);
    X(0, 0) = c; X(0, 1) = -s; X(1, 0) = s; X(1, 1) = c;
    X(0, 3) = x; X(1, 3) = y;
    data.push_back(X);
  }
  return data;
}

inline IsometryVector get_est_data(const std::string& file_path) {                  // evaluation_utils.cpp:32-64
  IsometryVector traj;
  std::ifstream in(file_path);
  if (!in.is_open()) { std::cout << "Unable to open " << file_path << std::endl; return traj; }
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    Isometry3f X = Isometry3f::Identity();
    ss >> X(0, 3) >> X(1, 3) >> X(2, 3);
    for (int i = 0; i < 3; i++) { std::getline(in, line); std::stringstream s2(line); for (int j = 0; j < 3; j++) s2 >> X(i, j); }
    traj.push_back(X);
  }
  return traj;
}

struct EvalResult {
  float mean_orientation_error = 0, median_ratio_inv = 0, rmse_position = 0, rmse_map = 0;
  int matched_map_points = 0;
};

//! evaluate.cpp:18-86; `out_performance` may be empty to skip the per-pair file
inline EvalResult evaluate(const IsometryVector& gt, const IsometryVector& est, const Vector3fVector& map_est,
                           const Vector10fVector& map_app, const Vector3fVector& world, const Vector10fVector& world_app,
                           const std::string& out_performance = "") {
  EvalResult r;
  std::vector<float> orientation_error, ratio;
  std::ofstream perf;
  if (!out_performance.empty()) perf.open(out_performance);
  auto tnorm = [](const Isometry3f& X) { return std::sqrt(X(0, 3) * X(0, 3) + X(1, 3) * X(1, 3) + X(2, 3) * X(2, 3)); };
  for (size_t i = 1; i < gt.size() && i < est.size(); i++) {
    const Isometry3f X_rel = est[i - 1].inverse() * est[i], X_rel_gt = gt[i - 1].inverse() * gt[i];
    float tr = 0.f;                                            // trace(I - R_rel^T R_rel_gt)
    for (int d = 0; d < 3; ++d) {
      float acc = 0.f;
      for (int k = 0; k < 3; ++k) acc += X_rel(k, d) * X_rel_gt(k, d);
      tr += 1.f - acc;
    }
    orientation_error.push_back(tr);
    ratio.push_back(tnorm(X_rel) / tnorm(X_rel_gt));
    if (perf.is_open()) perf << tr << " " << ratio.back() << "\n";
  }
  double mean = 0;
  for (float e : orientation_error) mean += e;
  r.mean_orientation_error = orientation_error.empty() ? 0.f : (float)(mean / orientation_error.size());
  r.median_ratio_inv = 1.f / median(ratio);
  double se = 0;
  const size_t n = std::min(gt.size(), est.size());
  for (size_t i = 0; i < n; i++)
    for (int d = 0; d < 3; ++d) { const float e = gt[i](d, 3) - est[i](d, 3) * r.median_ratio_inv; se += (double)e * e; }
  r.rmse_position = (float)std::sqrt(se / (double)n);
  // map: first landmark with exactly the same appearance (evaluate.cpp:71-80); hash instead of the O(M*W) scan
  auto key = [](const Vector10f& a) { return std::string(reinterpret_cast<const char*>(a.v), sizeof(a.v)); };
  std::unordered_map<std::string, size_t> first;
  for (size_t j = 0; j < world_app.size(); ++j) first.emplace(key(world_app[j]), j);
  double sm = 0;
  for (size_t i = 0; i < map_est.size() && i < map_app.size(); ++i) {
    auto it = first.find(key(map_app[i]));
    if (it == first.end()) continue;
    for (int d = 0; d < 3; ++d) { const float e = map_est[i][d] * r.median_ratio_inv - world[it->second][d]; sm += (double)e * e; }
    r.matched_map_points++;
  }
  r.rmse_map = r.matched_map_points ? (float)std::sqrt(sm / r.matched_map_points) : 0.f;
  return r;
}

}  // namespace vo
