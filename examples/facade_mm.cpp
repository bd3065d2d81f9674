// The reference driver's main loop (C++/examples/dist_pgo.cpp:446-531) written against the C++ facade
// include/dpgo_amd.hpp: read_g2o -> chordal init -> { iterate; communicate; update } with all nodes on GPU 0.
//   facade_mm <file.g2o> <num_nodes> <iters> [loss: trivial|huber|gm|welsch] [accelerated: 0|1]
//   facade_mm --info <file.g2o> <num_nodes>        (host only: partition sizes, no GPU needed)
// Prints "<iter>: <2F> <2|grad F|>" like the reference (dist_pgo.cpp:493-494).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "dpgo_amd.hpp"

int main(int argc, char **argv) {
  if (argc >= 4 && !strcmp(argv[1], "--info")) {
    auto g = DPGO::Graph::read_g2o(argv[2], atoi(argv[3]));
    printf("d %d poses %d edges %d nodes %d\n", g->d(), g->num_poses(), g->num_edges(), g->num_nodes());
    for (int a = 0; a < g->num_nodes(); a++) {
      int n[2], m[2];
      g->sizes(a, n, m);
      printf("node %d: n %d %d m %d %d offset %d\n", a, n[0], n[1], m[0], m[1], g->offset(a));
    }
    return 0;
  }
  if (argc < 4) {
    fprintf(stderr, "usage: %s <file.g2o> <num_nodes> <iters> [loss] [accelerated]\n", argv[0]);
    return 2;
  }
  const int num_nodes = atoi(argv[2]), iters = atoi(argv[3]);
  const std::string loss = argc > 4 ? argv[4] : "trivial";
  const bool acc = argc > 5 ? atoi(argv[5]) != 0 : true;
  const DPGO::Loss l = loss == "huber" ? DPGO::Loss::Huber : loss == "gm" ? DPGO::Loss::GemanMcClure
                       : loss == "welsch" ? DPGO::Loss::Welsch : DPGO::Loss::None;
  auto graph = DPGO::Graph::read_g2o(argv[1], num_nodes);
  std::vector<int> all(num_nodes);
  for (int a = 0; a < num_nodes; a++) all[a] = a;
  DPGO::DPGOHashGroup dpgo_hash(graph, all, DPGO::Options::driver(l, acc), 0);
  if (dpgo_hash.initialize(graph->chordal_initialization()) != 0 || dpgo_hash.update() != 0) return 1;
  auto report = [&](int it) {
    double F = 0, g2 = 0;
    for (int a = 0; a < num_nodes; a++) {
      const DPGO::DPGOResult r = dpgo_hash[a].results(false);
      F += r.fobj;
      g2 += r.gradFnorm * r.gradFnorm;
    }
    printf("%d: %.10e %.10e\n", it, 2 * F, 2 * std::sqrt(g2));
  };
  report(0);
  for (int it = 1; it <= iters; it++) {
    if (dpgo_hash.iterate() != 0 || dpgo_hash.communicate() != 0 || dpgo_hash.update() != 0) return 1;
    report(it);
  }
  return 0;
}
