// frame_check -- one frame of the vo_complete loop (vo_complete.cpp:150-179)
// through the C++ facade: match -> join -> X*model -> init + n x oneRound ->
// triangulate_points (point-cloud overload).  Reads a binary frame written by
// tests/test_gpu_facade.py, writes the results for the test to compare with the
// oracle.    usage: frame_check <in.bin> <out.bin>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "vo/vo.hpp"

using namespace vo;

template <class T>
static void rd(FILE* f, T* p, size_t n) { if (n && fread(p, sizeof(T), n, f) != n) { perror("read"); exit(3); } }
template <class T>
static void wr(FILE* f, const T* p, size_t n) { if (n && fwrite(p, sizeof(T), n, f) != n) { perror("write"); exit(3); } }

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: frame_check in.bin out.bin\n"); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int hdr[9]; float thr; Matrix3f k; Isometry3f X_prev;
  rd(f, hdr, 9); rd(f, &thr, 1); rd(f, k.m, 9); rd(f, X_prev.m, 16);
  const int rows = hdr[0], cols = hdr[1], z_near = hdr[2], z_far = hdr[3], n_ref = hdr[4], n_cur = hdr[5],
            n_model = hdr[6], n_mp = hdr[7], n_iters = hdr[8];
  PointCloudVector<2> reference_pc((size_t)n_ref), current_pc((size_t)n_cur);
  PointCloudVector<3> triangulated_pc((size_t)n_model);
  IntPairVector correspondences_world((size_t)n_mp);
  rd(f, reference_pc.points().data(), (size_t)n_ref); rd(f, reference_pc.appearances().data(), (size_t)n_ref);
  rd(f, current_pc.points().data(), (size_t)n_cur); rd(f, current_pc.appearances().data(), (size_t)n_cur);
  rd(f, triangulated_pc.points().data(), (size_t)n_model);
  rd(f, correspondences_world.data(), (size_t)n_mp);
  fclose(f);
  try {
    Camera cam(rows, cols, z_near, z_far, k);
    PICPSolver solver;
    solver.setKernelThreshold(thr);
    // vo_complete.cpp:156-173
    IntPairVector correspondences_imgs = compute_correspondences_images(reference_pc.appearances(), current_pc.appearances());
    correspondences_world = extract_correspondences_world(correspondences_imgs, correspondences_world);
    PointCloudVector<3> triangulated_transformed = X_prev * triangulated_pc;
    cam.setWorldInCameraPose(Isometry3f::Identity());
    solver.init(cam, triangulated_transformed.points(), current_pc.points());
    for (int i = 0; i < n_iters; i++) solver.oneRound(correspondences_world, false);
    cam = solver.camera();
    const float chi_in = solver.chiInliers(), chi_out = solver.chiOutliers();
    const int n_in = solver.numInliers();
    IntPairVector correspondences_new;
    PointCloudVector<3> tri_new;
    const int n_tri = triangulate_points(k, cam.worldInCameraPose(), correspondences_imgs, reference_pc, current_pc, tri_new,
                                         correspondences_new);
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    const int counts[4] = {(int)correspondences_imgs.size(), (int)correspondences_world.size(), n_tri, n_in};
    wr(o, counts, 4);
    wr(o, correspondences_imgs.data(), correspondences_imgs.size());
    wr(o, correspondences_world.data(), correspondences_world.size());
    wr(o, cam.worldInCameraPose().m, 16);
    wr(o, &chi_in, 1); wr(o, &chi_out, 1);
    wr(o, tri_new.points().data(), tri_new.size());
    wr(o, correspondences_new.data(), correspondences_new.size());
    wr(o, tri_new.appearances().data(), tri_new.size());
    wr(o, triangulated_transformed.points().data(), triangulated_transformed.size());
    fclose(o);
    std::printf("frame_check: %d matches, %d joined, %d inliers, %d triangulated\n", counts[0], counts[1], n_in, n_tri);
    if (argc > 3 && std::string(argv[3]) == "extras") {
      // TreeNode_ facade (eigen_kdtree.h): the approximate answers must be answers of the exact modes
      KdTree tree(reference_pc.appearances(), 10);
      const std::vector<int> full = tree.bestMatchFull(current_pc.appearances(), 0.1f), fast = tree.bestMatchFast(current_pc.appearances(), 0.1f);
      const std::vector<std::vector<int>> all = tree.fullSearch(current_pc.appearances(), 0.1f), some = tree.fastSearch(current_pc.appearances(), 0.1f);
      int same = 0, bad = 0;
      for (size_t i = 0; i < full.size(); ++i) {
        same += fast[i] == full[i];
        if (fast[i] >= 0 && std::find(all[i].begin(), all[i].end(), fast[i]) == all[i].end()) ++bad;
        for (int j : some[i]) if (std::find(all[i].begin(), all[i].end(), j) == all[i].end()) ++bad;
        if (full[i] >= 0 && std::find(all[i].begin(), all[i].end(), full[i]) == all[i].end()) ++bad;
      }
      std::printf("kdtree: bestMatchFast == bestMatchFull for %d of %zu queries, %d inconsistent answers\n", same, full.size(), bad);
      // PICPSolver is copyable like the reference's: a copy carries the settings, and after its own init() runs the same rounds
      PICPSolver copy = solver;
      cam.setWorldInCameraPose(Isometry3f::Identity());
      copy.init(cam, triangulated_transformed.points(), current_pc.points());
      for (int i = 0; i < n_iters; i++) copy.oneRound(correspondences_world, false);
      const Isometry3f a = copy.camera().worldInCameraPose(), b = solver.camera().worldInCameraPose();
      bool equal = copy.kernelThreshold() == solver.kernelThreshold();
      for (int i = 0; i < 16; ++i) equal = equal && a.m[i] == b.m[i];
      std::printf("solver copy: %s\n", equal ? "identical result" : "DIFFERENT result");
      if (bad || !equal) return 4;
    }
    return 0;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "frame_check: %s\n", e.what());
    return 2;
  }
}
