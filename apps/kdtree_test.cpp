// kdtree_test -- counterpart of the reference's src/tests/eigen_kdtree_test.cpp (which its CMake never builds: it includes a
// header that does not exist): random points, a tree with small leaves, queries of which about half are copies of tree points;
// for each query the approximate search (bestMatchFast: one side of every split, then the leaf) against the exact one
// (bestMatchFull).  The reference's test uses 2-D points; the tree of this library is the 10-D appearance tree the solver path
// uses, so the points live in the first two components of a 10-D vector.
//   usage: kdtree_test [seed=1] [n_points=50] [n_queries=10] [max_points_in_leaf=10]
// Prints like the reference ("FAST Correct" / "FAST Not Correct").  Approximate answers may legitimately miss; what must hold and
// decides the exit code: a fast hit is within the radius, an exact hit is never farther than the fast one, a query that copies a
// tree point is found by the exact search, and bestMatchFast == the closest member of fastSearch's list.
#include <cmath>

#include "synth.hpp"
#include "vo/vo.hpp"

using namespace vo;

int main(int argc, char** argv) {
  const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
  const int n_points = argc > 2 ? atoi(argv[2]) : 50, n_queries = argc > 3 ? atoi(argv[3]) : 10, leaf = argc > 4 ? atoi(argv[4]) : 10;
  try {
    synth::Rng rng(seed);
    Vector10fVector kd_points((size_t)n_points), query_points((size_t)n_queries);
    for (auto& p : kd_points) { for (int k = 0; k < 10; ++k) p[k] = 0.f; p[0] = rng.uniform(-10.f, 10.f); p[1] = rng.uniform(-10.f, 10.f); }
    std::vector<int> copy_of((size_t)n_queries, -1);
    for (int i = 0; i < n_queries; ++i) {
      auto& q = query_points[(size_t)i];
      for (int k = 0; k < 10; ++k) q[k] = 0.f;
      if (rng.uniform(0.f, 1.f) > 0.5f) { q[0] = rng.uniform(-10.f, 10.f); q[1] = rng.uniform(-10.f, 10.f); }
      else { copy_of[(size_t)i] = i % (n_points < 10 ? n_points : 10); q = kd_points[(size_t)copy_of[(size_t)i]]; }
    }
    const float ball_radius = 0.1f;
    KdTree kd_tree(kd_points, leaf);
    std::printf("tree ok: %d points, %d nodes, %d leaves\n", kd_tree.size(), kd_tree.nodes(), kd_tree.leaves());
    const std::vector<int> fast = kd_tree.bestMatchFast(query_points, ball_radius), full = kd_tree.bestMatchFull(query_points, ball_radius);
    const auto lists = kd_tree.fastSearch(query_points, ball_radius);
    auto d2 = [&](int qi, int ti) { double s = 0; for (int k = 0; k < 10; ++k) { const double d = (double)query_points[(size_t)qi][k] - kd_points[(size_t)ti][k]; s += d * d; } return s; };
    int bad = 0;
    for (int i = 0; i < n_queries; ++i) {
      const int mf = fast[(size_t)i], me = full[(size_t)i];
      if (mf == me) std::printf(mf >= 0 ? "FAST Correct: query %d -> point %d\n" : "FAST Correct, no match (query %d)\n", i, mf);
      else std::printf("FAST Not Correct: query %d fast %d full %d\n", i, mf, me);
      if (mf >= 0 && !(d2(i, mf) < (double)ball_radius * ball_radius)) { std::printf("  fast hit outside the radius\n"); ++bad; }
      if (mf >= 0 && (me < 0 || d2(i, me) > d2(i, mf))) { std::printf("  exact search worse than the approximate one\n"); ++bad; }
      if (copy_of[(size_t)i] >= 0 && (me < 0 || d2(i, me) != 0.0)) { std::printf("  a copied point was not found by the exact search\n"); ++bad; }
      int best = -1;
      for (int t : lists[(size_t)i]) if (best < 0 || d2(i, t) < d2(i, best) || (d2(i, t) == d2(i, best) && t < best)) best = t;
      if ((best < 0) != (mf < 0) || (best >= 0 && d2(i, best) != d2(i, mf))) { std::printf("  bestMatchFast is not the closest of fastSearch's list\n"); ++bad; }
    }
    return bad ? 1 : 0;
  } catch (const vo::Error& e) {
    std::fprintf(stderr, "kdtree_test: %s\n", e.what());
    return 2;
  }
}
