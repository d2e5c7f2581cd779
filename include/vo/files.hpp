// vo/files.hpp -- dataset I/O of the reference (files_utils.h / .cpp), same
// function names and file formats:
//   meas-XXXXX.dat : 3 header lines, then "point <k> <id> <col> <row> <a0..a9>"
//   world.dat      : "<id> <x> <y> <z> <a0..a9>"
//   camera.dat     : "camera matrix:" + 3 rows, "cam_transform:" + 4 rows,
//                    "z_near:", "z_far:", "width:", "height:"
//   trajectory.dat : "<k> <odom x y th> <gt x y th>"
// Cold host code: plain iostreams.
#pragma once

#include <dirent.h>

#include <fstream>
#include <iomanip>
#include <iostream>
#include <regex>
#include <set>
#include <sstream>
#include <string>

#include "point_cloud.hpp"
#include "types.hpp"

namespace vo {

//! file names in `path` matching `pattern`, alphabetical (files_utils.cpp:3-18)
inline bool get_file_names(const std::string& path, std::set<std::string>& files, const std::regex& pattern) {
  files.clear();
  DIR* dir = opendir(path.c_str());
  if (!dir) return false;
  while (dirent* ent = readdir(dir)) {
    const std::string name = ent->d_name;
    if (std::regex_search(name, pattern)) files.insert(name);
  }
  closedir(dir);
  return true;
}

//! id-keeping reader (files_utils.cpp:19-57): features = (id|x, col|y, row|z)
inline bool get_meas_content(const std::string& file_path, Vector10fVector& appearances, Vector3fVector& features,
                             const bool& is_world = false) {
  appearances.clear();
  features.clear();
  std::ifstream in(file_path);
  if (!in.is_open()) return false;
  std::string line, word;
  if (!is_world) for (int i = 0; i < 3; i++) std::getline(in, line);
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    ss >> word;                       // "point", or the id of world.dat
    if (!is_world) ss >> word;        // index in frame
    Vector3f f; Vector10f a; float n = 0.f;
    for (int i = 0; i < 13; i++) { ss >> n; if (i < 3) f[i] = n; else a[i - 3] = n; }
    features.push_back(f);
    appearances.push_back(a);
  }
  return true;
}

//! point-cloud reader (files_utils.cpp:58-93): (col,row) + appearance
inline bool get_meas_content(const std::string& file_path, PointCloudVector<2>& points) {
  points.clear();
  std::ifstream in(file_path);
  if (!in.is_open()) return false;
  std::string line, word;
  for (int i = 0; i < 3; i++) std::getline(in, line);
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    ss >> word >> word >> word;       // "point" k id
    Vector2f f; Vector10f a; float n = 0.f;
    for (int i = 0; i < 12; i++) { ss >> n; if (i < 2) f[i] = n; else a[i - 2] = n; }
    points.push_back(PointCloud<2>(f, a));
  }
  return true;
}

//! files_utils.cpp:94-134: int_params = z_near, z_far, width, height (file order)
inline bool get_camera_params(const std::string& file_path, std::vector<int>& int_params, Matrix3f& k, Isometry3f& H) {
  std::ifstream in(file_path);
  if (!in.is_open()) return false;
  int_params.clear();
  H = Isometry3f::Identity();
  std::string line, keyword;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    ss >> keyword;
    if (keyword == "camera") {
      for (int i = 0; i < 3; i++) { std::getline(in, line); std::stringstream s2(line); for (int j = 0; j < 3; j++) s2 >> k(i, j); }
    } else if (keyword == "cam_transform:") {
      for (int i = 0; i < 4; i++) { std::getline(in, line); std::stringstream s2(line); for (int j = 0; j < 4; j++) s2 >> H(i, j); }
    } else if (keyword == "z_near:" || keyword == "z_far:" || keyword == "width:" || keyword == "height:") {
      int n = 0; ss >> n; int_params.push_back(n);
    }
  }
  return true;
}

//! one vector per line (files_utils.h:17-28)
template <class Vec>
inline void write_eigen_vectors_to_file(const std::string& file_path, const std::vector<Vec>& vectors) {
  std::ofstream out(file_path);
  if (!out.is_open()) { std::cout << "Error opening file" << std::endl; return; }
  out << std::setprecision(9);
  for (const auto& v : vectors) {
    for (size_t i = 0; i < sizeof(Vec) / sizeof(float); ++i) out << (i ? " " : "") << v[(int)i];
    out << "\n";
  }
}

//! files_utils.cpp:136-153: H <- H * C * X_i^-1 * C^-1, the i-th robot pose in the world
inline void save_trajectory(const std::string& file_path, const IsometryVector& vector,
                            const Isometry3f& cameraInRobot = Isometry3f::Identity(), const bool& save_rotation = false) {
  std::ofstream out(file_path);
  if (!out.is_open()) { std::cout << "Unable to open " << file_path << " where to save the trajectory" << std::endl; return; }
  out << std::setprecision(9);
  Isometry3f H = Isometry3f::Identity();
  const Isometry3f Ci = cameraInRobot.inverse();
  for (const auto& X : vector) {
    H = H * cameraInRobot * X.inverse() * Ci;
    out << H(0, 3) << " " << H(1, 3) << " " << H(2, 3) << "\n";
    if (save_rotation) for (int r = 0; r < 3; ++r) out << H(r, 0) << " " << H(r, 1) << " " << H(r, 2) << "\n";
  }
}

//! files_utils.cpp:155-182: ground-truth (x, y, 0) per line -> trajectory_gt.txt
inline bool save_gt_trajectory(const std::string& file_path, const std::string& out_path = "trajectory_gt.txt") {
  std::ifstream in(file_path);
  if (!in.is_open()) { std::cout << "Unable to open " << file_path << std::endl; return false; }
  Vector3fVector pts;
  std::string line, word;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::stringstream ss(line);
    for (int i = 0; i < 4; i++) ss >> word;
    Vector3f p = Vector3f::Zero();
    ss >> p[0] >> p[1];
    pts.push_back(p);
  }
  write_eigen_vectors_to_file(out_path, pts);
  return true;
}

}  // namespace vo
