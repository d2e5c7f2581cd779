// compute_corr -- counterpart of the reference's src/tests/compute_corr.cpp:84-119 on the GPU path: the appearance
// matcher (compute_correspondences_images) against the association the landmark ids give, for every consecutive pair of
// measurement files of a data directory (the reference prints the first pair of a hard-coded path).
//   usage: compute_corr <data dir> [--verbose]
// exit code 0 iff every pair of every frame agrees.
#include "known_common.hpp"

using namespace vo;
using namespace known;

int main(int argc, char* argv[]) {
  if (argc < 2) { std::cout << "Error: need path parameter to read data" << std::endl; return -1; }
  std::string path(argv[1]);
  if (path.back() != '/') path.push_back('/');
  const bool verbose = argc > 2 && std::string(argv[2]) == "--verbose";
  try {
    const std::regex pattern("^meas-\\d.*\\.dat$");
    std::set<std::string> files;
    if (!get_file_names(path, files, pattern)) { std::cout << "unable to open directory\n"; return -1; }
    Vector3fVector reference_image_points_withid, current_image_points_withid;   // (landmark id, col, row)
    Vector10fVector reference_appearances, current_appearances;
    std::string previous;
    size_t frames = 0, pairs = 0, wrong = 0;
    for (const auto& file : files) {
      if (!get_meas_content(path + file, current_appearances, current_image_points_withid)) { std::cout << "Unable to open file " << path + file << std::endl; return -1; }
      if (!previous.empty()) {
        // the pairs are (ref_idx,curr_idx)
        const IntPairVector correspondences_imgs = compute_correspondences_images(reference_appearances, current_appearances);
        const IntPairVector correspondences_imgs_gt = extract_correspondences_images(reference_image_points_withid, current_image_points_withid);
        const bool same = correspondences_imgs == correspondences_imgs_gt;
        if (verbose || !same) {
          std::printf("%s -> %s sizes: %zu, %zu%s\n", previous.c_str(), file.c_str(), correspondences_imgs_gt.size(), correspondences_imgs.size(), same ? "" : "  DIFFERENT");
          for (size_t i = 0; verbose && i < correspondences_imgs_gt.size() && i < correspondences_imgs.size(); i++)
            std::printf("gt: %d, %d  est: %d, %d\n", correspondences_imgs_gt[i].first, correspondences_imgs_gt[i].second, correspondences_imgs[i].first, correspondences_imgs[i].second);
        }
        ++frames; pairs += correspondences_imgs_gt.size(); wrong += same ? 0 : 1;
      }
      previous = file;
      reference_image_points_withid = current_image_points_withid;
      reference_appearances = current_appearances;
    }
    std::printf("%zu consecutive frame pairs, %zu correspondences: appearance matcher %s the id association%s\n", frames, pairs,
                wrong ? "DIFFERS FROM" : "equals", wrong ? "" : " everywhere");
    return wrong ? 1 : 0;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "compute_corr: %s\n", e.what());
    return 2;
  }
}
