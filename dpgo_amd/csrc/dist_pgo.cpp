// dist_pgo -- command-line driver with the reference's flags and outputs, on top of the C ABI.
//
// Mirrors C++/examples/dist_pgo.cpp:
//   flags      --dataset --num_nodes --iters[1000] --dist_init[true] --loss[trivial|huber|welsch]
//              --accelerated[true] --save[true]                                     (:23-47)
//   options    the hard-coded overrides of :103-120 (dpgo_options_driver)
//   loop       iterate -> gather -> communicate -> update, timing iterate + update only (:492-531)
//   stdout     "<iter>: <fobj> <grad>" with 20 digits, then the final summary         (:493-494, 533-536)
//   files      results_chordal_<N>_<amm|mm>.txt  (iter time fobj grad, 16 digits)     (:538-552)
//              estimates_<loss>.txt  (X with t <- t - t_0, X <- X R_0)                (:554-567)
// One process hosts all nodes on one GPU (--gpu), or -- one process per GPU -- rank r of n (--rank / --world, or
// RANK / WORLD_SIZE / LOCAL_RANK from a launcher) hosts nodes [r num_nodes / n, (r + 1) num_nodes / n) and the
// boundary poses travel by RCCL (dpgo_comm_exchange); the 128-byte RCCL id goes from rank 0 to the others through
// the private directory --rdv (default: named after the launcher's run id / MASTER_PORT) with a nonce handshake.  fobj = 2 F and grad = 2 |grad F| come from the per-node
// device reductions (sum_a fobj^a = F, sum_a |Proj(Dfobj^a)|^2 = |grad F|^2; DPGOStar.cpp:713-829), summed over
// the ranks.
// --dist_init true runs the distributed chordal initialisation (:144-416, C++/DChordal) through
// dpgo_group_dist_chordal_initialization and prints the stage objectives the reference prints every 20
// iterations (:206-210); its stage 0 (a per-node SE-Sync solve in the reference) is the library's stand-in.
// --dist_init false is the centralised chordal initialisation (:416-444).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "../../include/dpgo_amd.h"
#include "rdv.h"


static bool parse_bool(const char *s) { return !(strcmp(s, "false") == 0 || strcmp(s, "0") == 0); }

int main(int argc, char **argv) {
  std::string dataset, loss_type = "trivial";
  int num_nodes = -1, iters = 1000, gpu = -1;
  bool dist_init = true, accelerated = true, save = true;
  int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0, world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
  std::string rdv;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto val = [&](const char *name) -> const char * {
      const size_t n = strlen(name);
      if (a.compare(0, n, name) == 0 && a.size() > n && a[n] == '=') return argv[i] + n + 1;
      if (a == name && i + 1 < argc) return argv[++i];
      return nullptr;
    };
    if (a == "--help") {
      printf("Program options:\n  --dataset arg\n  --num_nodes arg\n  --iters arg (=1000)\n  --dist_init arg (=true)\n"
             "  --loss arg (=trivial)   trivial, huber or welsch\n  --accelerated arg (=true)\n  --save arg (=true)\n  --gpu arg (=0)\n");
      return 0;
    } else if (const char *v = val("--dataset")) dataset = v;
    else if (const char *v = val("--num_nodes")) num_nodes = atoi(v);
    else if (const char *v = val("--iters")) iters = atoi(v);
    else if (const char *v = val("--dist_init")) dist_init = parse_bool(v);
    else if (const char *v = val("--loss")) loss_type = v;
    else if (const char *v = val("--accelerated")) accelerated = parse_bool(v);
    else if (const char *v = val("--save")) save = parse_bool(v);
    else if (const char *v = val("--gpu")) gpu = atoi(v);
    else if (const char *v = val("--rank")) rank = atoi(v);
    else if (const char *v = val("--world")) world = atoi(v);
    else if (const char *v = val("--rdv")) rdv = v;
  }
  if (gpu < 0) gpu = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : (world > 1 ? rank : 0);
  if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "Inconsistent --rank / --world.\n"); return -1; }
  if (world > 1 && rdv.empty()) {
    // a name all ranks of this launch -- and no other launch -- agree on: the launcher's run id / port
    const char *run = getenv("TORCHELASTIC_RUN_ID"), *port = getenv("MASTER_PORT");
    if (!run && !port) { fprintf(stderr, "Several ranks need --rdv <directory> (or a launcher that sets MASTER_PORT).\n"); return -1; }
    rdv = std::string("/tmp/dpgo_rdv_") + std::to_string((long)geteuid()) + "_" + (run ? run : "none") + "_" + (port ? port : "0");
  }
  if (dataset.empty()) { fprintf(stderr, "No dataset has been specfied.\n"); return -1; }
  if (num_nodes < 1) { fprintf(stderr, "No number of nodes has been specfied.\n"); return -1; }
  int loss;
  if (loss_type == "trivial") loss = 0;
  else if (loss_type == "huber") loss = 1;
  else if (loss_type == "welsch") loss = 3;
  else { fprintf(stderr, " The loss type can only be \"trivial\", \"huber\" or \"welsch\".\n"); return -1; }

  dpgo_graph_t *g = nullptr;
  if (dpgo_read_g2o(dataset.c_str(), num_nodes, &g) != 0) return -1;
  int d, N, nn, m;
  dpgo_graph_info(g, &d, &N, &nn, &m);
  dpgo_options_t opt;
  dpgo_options_driver(&opt, loss, accelerated);
  const int ld = (d + 1) * N;
  std::vector<double> X((size_t)ld * d, 0.0);
  if (num_nodes % world != 0) { fprintf(stderr, "The number of nodes must be a multiple of the number of ranks.\n"); return -1; }
  const int per = num_nodes / world;
  std::vector<int> ids(per);
  for (int a = 0; a < per; a++) ids[a] = rank * per + a;
  const bool root = rank == 0;
  dpgo_group_t *grp = nullptr;
  if (dpgo_group_create(g, ids.data(), per, &opt, gpu, &grp) != 0) return -1;
  dpgo_comm_t *comm = nullptr;
  if (world > 1 || getenv("DPGO_FORCE_COMM")) {   // (DPGO_FORCE_COMM: a world of one rank still runs the whole protocol)
    unsigned char id[128];
    if (root && dpgo_comm_unique_id(id) != 0) return -1;
    if (world > 1 && dpgo_rdv::rendezvous(rdv, rank, world, id) != 0) return -1;
    if (dpgo_comm_create(grp, rank, world, id, &comm) != 0) return -1;
    dpgo_comm_barrier(comm);
  }
  if (dist_init) {
    // every node of the graph takes part in the distributed initialisation; with several ranks each runs the schedule
    // for its own nodes and the stage halos travel through the communicator (dchordal.cpp)
    dpgo_dchordal_options_t co;
    dpgo_dchordal_options_default(&co);
    int cap = 0;
    for (int k = 0; k < 4; k++) cap += (co.iters[k] + 19) / 20;
    std::vector<double> obj(cap);
    if (dpgo_group_dist_chordal_initialization(grp, &co, nullptr, 0, X.data(), ld, obj.data(), &cap) != 0) return -1;
    const char *names[4] = {"Initialize the reduced rotation", "Initialize the rotation", "Initialize the reduced translation",
                            "Initialize the translation"};
    int at = 0;
    for (int k = 0; k < 4 && root; k++) {
      printf("===============================================\n%s\n-----------------------------------------------\n", names[k]);
      for (int it = 0; it < co.iters[k]; it += 20) printf("%d: %.16g\n", it, obj[at++]);
    }
  } else {
    if (root) printf("===============================================\nInitialization\n-----------------------------------------------\n");
    if (dpgo_chordal_initialization(g, X.data(), ld) != 0) return -1;
  }
  if (dpgo_group_initialize_global(grp, X.data(), ld) != 0) return -1;
  if (dpgo_group_update(grp, nullptr, 0) != 0) return -1;

  auto evaluate = [&](double &fobj, double &grad) {
    double v[2] = {0, 0};
    for (int a = 0; a < per; a++) {
      dpgo_results_t r;
      dpgo_group_results(grp, a, &r);
      v[0] += r.fobj;
      v[1] += r.gradFnorm * r.gradFnorm;
    }
    if (comm) dpgo_comm_allreduce_sum(comm, v, 2);
    fobj = 2 * v[0];
    grad = 2 * std::sqrt(v[1]);
  };
  double fobj, grad, time = 0;
  evaluate(fobj, grad);
  std::vector<std::vector<double>> results;
  results.push_back({0, 0, fobj, grad});
  if (root) printf("===============================================\nDistributed PGO\n-----------------------------------------------\n");
  using clk = std::chrono::steady_clock;
  for (int iter = 0; iter < iters; iter++) {
    if (root) printf("%d: %.20g %.20g\n", iter, fobj, grad);
    auto t0 = clk::now();
    if (dpgo_group_iterate(grp, nullptr, 0) != 0) return -1;
    dpgo_group_sync(grp);
    time += std::chrono::duration<double>(clk::now() - t0).count();
    if (comm && dpgo_comm_exchange(comm) != 0) return -1;   // neighbours on other GPUs: RCCL, joined by update()
    dpgo_group_communicate_local(grp);
    t0 = clk::now();
    if (dpgo_group_update(grp, nullptr, 0) != 0) return -1;
    dpgo_group_sync(grp);
    time += std::chrono::duration<double>(clk::now() - t0).count();
    evaluate(fobj, grad);
    results.push_back({double(iter) + 1, time, fobj, grad});
  }
  if (root)
    printf("---------------------------------------\nfinal objective: %.20g\nfinal gradient: %.20g\ntime: %.20g s/node.\n", fobj,
           grad, time / per);
  if (save) {
    std::fill(X.begin(), X.end(), 0.0);
    dpgo_group_scatter_global(grp, X.data(), ld);
    if (comm) dpgo_comm_allreduce_sum(comm, X.data(), (long)X.size());   // every rank contributes its own poses
  }
  if (save && root) {
    const std::string resfile = "results_chordal_" + std::to_string(num_nodes) + "_" + (accelerated ? "amm" : "mm") + ".txt";
    FILE *f = fopen(resfile.c_str(), "w");
    if (!f) return -1;
    for (const auto &r : results) fprintf(f, "%d %.16g %.16g %.16g\n", (int)r[0], r[1], r[2], r[3]);
    fclose(f);
    // gauge fix (:554-558): t <- t - t_0 ; X <- X * R with R = X.middleRows(num_poses, d)^T
    std::vector<double> R(d * d), t0v(d);
    for (int c = 0; c < d; c++) {
      t0v[c] = X[(size_t)c * ld + 0];
      for (int r = 0; r < d; r++) R[r * d + c] = X[(size_t)r * ld + N + c];   // R(r,c) = block(c,r)
    }
    for (int i = 0; i < N; i++)
      for (int c = 0; c < d; c++) X[(size_t)c * ld + i] -= t0v[c];
    f = fopen(("./estimates_" + loss_type + ".txt").c_str(), "w");
    if (!f) return -1;
    std::vector<double> row(d);
    for (int i = 0; i < ld; i++) {
      for (int c = 0; c < d; c++) {
        double s = 0;
        for (int k = 0; k < d; k++) s += X[(size_t)k * ld + i] * R[k * d + c];
        row[c] = s;
      }
      for (int c = 0; c < d; c++) fprintf(f, "%s%g", c ? " " : "", row[c]);
      fprintf(f, "\n");
    }
    fclose(f);
  }
  if (comm) dpgo_comm_free(comm);
  dpgo_group_free(grp);
  dpgo_graph_free(g);
  return 0;
}
