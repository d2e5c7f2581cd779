// Shared by the known-association drivers (counterparts of the reference's src/tests/picp_real_data_allKnown.cpp,
// vo_daKnown.cpp and initialization_real_data.cpp): the helpers those programs define in their own translation units.
#pragma once
#include <cmath>
#include <cstdio>
#include <regex>
#include <set>
#include <string>
#include <vector>

#include "vo/vo.hpp"

namespace known {
using namespace vo;

// (id, col, row) -> (col, row)          (picp_real_data_allKnown.cpp:9-16)
inline Vector2fVector strip_id(const Vector3fVector& p_withid) {
  Vector2fVector ret;
  ret.reserve(p_withid.size());
  for (const auto& el : p_withid) ret.push_back(Vector2f{{el[1], el[2]}});
  return ret;
}

// (measurement index, landmark id)      (picp_real_data_allKnown.cpp:18-24)
inline IntPairVector computeFakeCorrespondencesWorld(const Vector3fVector& image_points_withid) {
  IntPairVector correspondences(image_points_withid.size());
  for (size_t i = 0; i < image_points_withid.size(); i++) correspondences[i] = IntPair((int)i, (int)image_points_withid[i].x());
  return correspondences;
}

// (ref_idx, curr_idx) of equal landmark id; the measurements are ordered by id     (vo_daKnown.cpp:20-35)
inline IntPairVector extract_correspondences_images(const Vector3fVector& reference_image_points_withid,
                                                    const Vector3fVector& current_image_points_withid) {
  IntPairVector correspondences;
  correspondences.reserve(current_image_points_withid.size());
  for (size_t i = 0; i < reference_image_points_withid.size(); i++) {
    for (size_t j = 0; j < current_image_points_withid.size(); j++) {
      if (current_image_points_withid[j].x() > reference_image_points_withid[i].x()) break;
      if (current_image_points_withid[j].x() == reference_image_points_withid[i].x()) {
        correspondences.push_back(IntPair((int)i, (int)j));
        break;
      }
    }
  }
  return correspondences;
}

// one camera pose per line, row-major 4x4, %.9g = exact float32 round trip (as vo_complete's poses_raw.txt)
inline void write_poses_raw(const std::string& file, const IsometryVector& trajectory) {
  std::FILE* f = std::fopen(file.c_str(), "w");
  if (!f) return;
  for (const auto& X : trajectory) {
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) std::fprintf(f, "%.9g ", X(r, c));
    std::fprintf(f, "\n");
  }
  std::fclose(f);
}

// robot poses of the estimate (save_trajectory's composition, files_utils.cpp:136-153) against trajectory.dat:
// largest absolute difference of any matrix entry over the whole trajectory, translations scaled by `scale` first
inline float max_error_vs_gt(const IsometryVector& trajectory, const Isometry3f& cameraInRobot, const IsometryVector& gt, float scale = 1.f) {
  Isometry3f H = Isometry3f::Identity();
  const Isometry3f Ci = cameraInRobot.inverse();
  float err = 0.f;
  for (size_t i = 0; i < trajectory.size() && i < gt.size(); ++i) {
    H = H * cameraInRobot * trajectory[i].inverse() * Ci;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) err = std::fmax(err, std::fabs(H(r, c) * (c == 3 ? scale : 1.f) - gt[i](r, c)));
  }
  return err;
}

// Everything the known-association drivers read before their frame loop: the measurement file names (sorted), the camera
// (files.hpp readers), and -- when asked for -- the landmark map of world.dat.  Prints the reason and returns false when a
// file is missing.
struct Dataset {
  std::vector<std::string> meas_files;
  Matrix3f K;
  Isometry3f cameraInRobot;
  int rows = 0, cols = 0, z_near = 0, z_far = 0;
  Vector3fVector landmarks;            // world.dat positions, indexed by landmark id
  Camera camera() const { return Camera(rows, cols, z_near, z_far, K); }
};

inline bool load_dataset(const std::string& dir, Dataset& d, bool with_landmarks) {
  std::set<std::string> names;
  if (!get_file_names(dir, names, std::regex("^meas-\\d.*\\.dat$"))) { std::printf("unable to open directory %s\n", dir.c_str()); return false; }
  d.meas_files.assign(names.begin(), names.end());
  std::vector<int> ints;               // z_near, z_far, width, height (files_utils.cpp:94-134)
  if (!get_camera_params(dir + "camera.dat", ints, d.K, d.cameraInRobot) || ints.size() < 4) { std::printf("unable to read %scamera.dat\n", dir.c_str()); return false; }
  d.z_near = ints[0]; d.z_far = ints[1]; d.cols = ints[2]; d.rows = ints[3];
  if (with_landmarks) {
    Vector10fVector unused_appearances;
    if (!get_meas_content(dir + "world.dat", unused_appearances, d.landmarks, true)) { std::printf("unable to read %sworld.dat\n", dir.c_str()); return false; }
  }
  return true;
}

struct Args {
  std::string path, out = "./";
  int rounds = 1000;
  bool exact = false, ok = false;
};

// <data dir> [output dir] [rounds] [--exact]
inline Args parse(int argc, char** argv, int default_rounds) {
  Args a;
  a.rounds = default_rounds;
  std::vector<std::string> pos;
  for (int i = 1; i < argc; ++i) {
    const std::string s(argv[i]);
    if (s == "--exact") a.exact = true;
    else if (s.rfind("--", 0) == 0) { std::printf("unknown option %s\n", s.c_str()); return a; }
    else pos.push_back(s);
  }
  if (pos.empty()) { std::printf("Error: need path parameter to read data\n"); return a; }
  a.path = pos[0];
  if (a.path.back() != '/') a.path.push_back('/');
  if (pos.size() > 1) a.out = pos[1];
  if (a.out.back() != '/') a.out.push_back('/');
  if (pos.size() > 2) a.rounds = std::atoi(pos[2].c_str());
  a.ok = true;
  return a;
}
}  // namespace known
