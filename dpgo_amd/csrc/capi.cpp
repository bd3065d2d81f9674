// C ABI (include/dpgo_amd.h) over the C++ host layer.
#include "../../include/dpgo_amd.h"

#include <memory>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <numeric>
#include <set>

#include "comm.h"
#include "group.h"

namespace dpgo {
int chordal_initialization(const Graph &g, double *X, int ld);
}

// No exception leaves the library: a failed HIP call (dpgo::DeviceError) or an allocation failure becomes the
// reference's `return -1` (+ a line on stderr), never abort() / terminate() in the host process.
template <class F>
static int guarded(F &&f) {
  try {
    return f();
  } catch (const std::exception &e) {
    fprintf(stderr, "[dpgo_amd] ERROR: %s\n", e.what());
    return -1;
  } catch (...) {
    fprintf(stderr, "[dpgo_amd] ERROR: unknown exception\n");
    return -1;
  }
}

struct dpgo_graph {
  dpgo::Graph g;
};
struct dpgo_group {
  dpgo::Group *grp = nullptr;
  std::vector<int> all;
};
struct dpgo_comm {
  dpgo::Comm *c = nullptr;
};

static dpgo::Options to_cpp(const dpgo_options_t &o) {
  dpgo::Options r;
  r.scheme = o.scheme; r.regularizer = o.regularizer; r.accepted_delta = o.accepted_delta;
  r.eta[0] = o.eta[0]; r.eta[1] = o.eta[1]; r.psi = o.psi; r.phi = o.phi;
  r.max_soft_restart_hits[0] = o.max_soft_restart_hits[0]; r.max_soft_restart_hits[1] = o.max_soft_restart_hits[1];
  r.oscillation_cnt_period = o.oscillation_cnt_period; r.max_oscillations = o.max_oscillations;
  r.loss = o.loss; r.loss_reg = o.loss_reg; r.rescale = o.rescale; r.max_rescale_count = o.max_rescale_count;
  r.grad_norm_tol = o.grad_norm_tol;
  r.rel_func_decrease_tol = o.rel_func_decrease_tol; r.stepsize_tol = o.stepsize_tol;
  r.max_iterations = o.max_iterations; r.max_iterations_accepted = o.max_iterations_accepted;
  r.reg_Cholesky_precon_max_condition_number = o.reg_Cholesky_precon_max_condition_number;
  r.preconditioned_grad_norm_tol = o.preconditioned_grad_norm_tol; r.max_tCG_iterations = o.max_tCG_iterations;
  r.STPCG_kappa = o.STPCG_kappa; r.STPCG_theta = o.STPCG_theta; r.preconditioner = o.preconditioner;
  r.verbose = o.verbose;
  return r;
}

extern "C" {

void dpgo_options_default(dpgo_options_t *o) {
  const dpgo::Options d;
  o->scheme = d.scheme; o->regularizer = d.regularizer; o->accepted_delta = d.accepted_delta;
  o->eta[0] = d.eta[0]; o->eta[1] = d.eta[1]; o->psi = d.psi; o->phi = d.phi;
  o->max_soft_restart_hits[0] = d.max_soft_restart_hits[0]; o->max_soft_restart_hits[1] = d.max_soft_restart_hits[1];
  o->oscillation_cnt_period = d.oscillation_cnt_period; o->max_oscillations = d.max_oscillations;
  o->loss = d.loss; o->loss_reg = d.loss_reg; o->rescale = d.rescale; o->max_rescale_count = d.max_rescale_count;
  o->grad_norm_tol = d.grad_norm_tol;
  o->rel_func_decrease_tol = d.rel_func_decrease_tol; o->stepsize_tol = d.stepsize_tol;
  o->max_iterations = d.max_iterations; o->max_iterations_accepted = d.max_iterations_accepted;
  o->reg_Cholesky_precon_max_condition_number = d.reg_Cholesky_precon_max_condition_number;
  o->preconditioned_grad_norm_tol = d.preconditioned_grad_norm_tol; o->max_tCG_iterations = d.max_tCG_iterations;
  o->STPCG_kappa = d.STPCG_kappa; o->STPCG_theta = d.STPCG_theta; o->preconditioner = d.preconditioner;
  o->verbose = d.verbose;
}

void dpgo_options_driver(dpgo_options_t *o, int loss, int accelerated) {
  // C++/examples/dist_pgo.cpp:103-120
  dpgo_options_default(o);
  o->loss = loss;
  o->rescale = 0;   // Rescale::Static (dist_pgo.cpp:105)
  o->loss_reg = 0.25;
  o->scheme = accelerated ? 1 : 0;
  o->STPCG_kappa = 0.05;
  o->STPCG_theta = 0.9;
  o->eta[0] = 5e-4; o->eta[1] = 2.5e-2;
  o->max_soft_restart_hits[0] = 10; o->max_soft_restart_hits[1] = 25;
  o->max_iterations = 10;
  o->max_iterations_accepted = 1;
  o->grad_norm_tol = 1e-3;
  o->preconditioned_grad_norm_tol = 1e-4;
  o->regularizer = 1e-11;
}

int dpgo_read_g2o(const char *filename, int num_nodes, dpgo_graph_t **out) {
  auto *g = new dpgo_graph();
  if (dpgo::read_g2o(filename, num_nodes, g->g) != 0) { delete g; *out = nullptr; return -1; }
  *out = g;
  return 0;
}

int dpgo_graph_from_edges(int d, int num_poses, int m, const int *I, const int *J, const double *R, const double *t,
                          const double *kappa, const double *tau, int num_nodes, dpgo_graph_t **out) {
  *out = nullptr;
  if ((d != 2 && d != 3) || m <= 0 || num_poses <= 0 || num_nodes < 1 || !I || !J || !R || !t || !kappa || !tau) return -1;
  for (int e = 0; e < m; e++)
    if (I[e] < 0 || I[e] >= num_poses || J[e] < 0 || J[e] >= num_poses) {
      fprintf(stderr, "[dpgo_amd] ERROR: edge %d: pose id out of range (%d, %d), num_poses = %d.\n", e, I[e], J[e], num_poses);
      return -1;
    }
  auto *g = new dpgo_graph();
  g->g.d = d;
  g->g.num_poses = num_poses;
  g->g.all.resize(m);
  for (int e = 0; e < m; e++) {
    dpgo::Measurement &mm = g->g.all[e];
    std::memset(&mm, 0, sizeof(mm));
    mm.ipose = I[e]; mm.jpose = J[e];
    for (int k = 0; k < d * d; k++) mm.R[k] = R[(size_t)e * d * d + k];
    for (int k = 0; k < d; k++) mm.t[k] = t[(size_t)e * d + k];
    mm.kappa = kappa[e]; mm.tau = tau[e];
  }
  if (dpgo::partition(g->g, num_nodes) != 0) { delete g; *out = nullptr; return -1; }
  *out = g;
  return 0;
}

void dpgo_graph_free(dpgo_graph_t *g) { delete g; }

int dpgo_graph_info(const dpgo_graph_t *g, int *d, int *num_poses, int *num_nodes, int *num_edges) {
  if (!g) return -1;
  if (d) *d = g->g.d;
  if (num_poses) *num_poses = g->g.num_poses;
  if (num_nodes) *num_nodes = g->g.num_nodes;
  if (num_edges) *num_edges = (int)g->g.all.size();
  return 0;
}

int dpgo_graph_edges(const dpgo_graph_t *g, int *I, int *J, double *R, double *t, double *kappa, double *tau) {
  if (!g) return -1;
  const int d = g->g.d;
  for (size_t e = 0; e < g->g.all.size(); e++) {
    const auto &m = g->g.all[e];
    if (I) I[e] = m.ipose;
    if (J) J[e] = m.jpose;
    if (R) for (int k = 0; k < d * d; k++) R[e * d * d + k] = m.R[k];
    if (t) for (int k = 0; k < d; k++) t[e * d + k] = m.t[k];
    if (kappa) kappa[e] = m.kappa;
    if (tau) tau[e] = m.tau;
  }
  return 0;
}

static int node_info(const dpgo_graph_t *g, int node, dpgo::DataInfo &info) {
  if (!g || node < 0 || node >= g->g.num_nodes) return -1;
  return dpgo::generate_data_info(node, g->g.d, g->g.measurements[node], info);
}

int dpgo_graph_node_sizes(const dpgo_graph_t *g, int node, int *n0, int *n1, int *m0, int *m1) {
  dpgo::DataInfo info;
  if (node_info(g, node, info) != 0) return -1;
  if (n0) *n0 = info.n[0];
  if (n1) *n1 = info.n[1];
  if (m0) *m0 = info.m[0];
  if (m1) *m1 = info.m[1];
  return 0;
}

int dpgo_graph_node_neighbours(const dpgo_graph_t *g, int node, int *nbr_node, int *nbr_pose) {
  dpgo::DataInfo info;
  if (node_info(g, node, info) != 0) return -1;
  for (int k = 0; k < info.n[1]; k++) {
    nbr_node[k] = info.nbr_key[k].first;
    nbr_pose[k] = info.nbr_key[k].second;
  }
  return 0;
}

int dpgo_graph_node_offset(const dpgo_graph_t *g, int node) {
  if (!g || node < 0 || node >= g->g.num_nodes || g->g.g_index[node].empty()) return -1;
  return g->g.g_index[node].begin()->second;
}

int dpgo_graph_exchange_plan(const dpgo_graph_t *g, const int *node_ids, int num_local, int *sent_nodes,
                             int *sent_poses, int *recv_nodes, int *recv_poses, int *counts) {
  // which own poses the nodes of this group export to nodes outside it, and which neighbour poses
  // they import from outside (sorted by (node, pose)); pointers may be NULL to query the counts
  if (!g || num_local <= 0) return -1;
  std::set<int> local(node_ids, node_ids + num_local);
  std::set<std::pair<int, int>> sent, recv;
  for (int k = 0; k < num_local; k++) {
    dpgo::DataInfo info;
    if (node_info(g, node_ids[k], info) != 0) return -1;
    for (const auto &s : info.sent)
      if (!local.count(s.first))
        for (int row : s.second) sent.insert({node_ids[k], info.own_pose[row]});
    for (const auto &key : info.nbr_key)
      if (!local.count(key.first)) recv.insert(key);
  }
  counts[0] = (int)sent.size();
  counts[1] = (int)recv.size();
  int i = 0;
  if (sent_nodes) for (const auto &s : sent) { sent_nodes[i] = s.first; sent_poses[i] = s.second; i++; }
  i = 0;
  if (recv_nodes) for (const auto &s : recv) { recv_nodes[i] = s.first; recv_poses[i] = s.second; i++; }
  return 0;
}

int dpgo_chordal_initialization(const dpgo_graph_t *g, double *X, int ld) {
  if (!g) return -1;
  return dpgo::chordal_initialization(g->g, X, ld);
}

int dpgo_group_create(const dpgo_graph_t *g, const int *node_ids, int num_local, const dpgo_options_t *opt,
                      int device, dpgo_group_t **out) {
  *out = nullptr;
  if (!g || num_local <= 0) return -1;
  if (!node_ids || !opt) return -1;
  std::vector<int> ids(node_ids, node_ids + num_local);
  if (std::set<int>(ids.begin(), ids.end()).size() != ids.size()) {
    fprintf(stderr, "[dpgo_amd] ERROR: dpgo_group_create: a node id appears twice.\n");
    return -1;
  }
  for (int id : ids)
    if (id < 0 || id >= g->g.num_nodes) {
      fprintf(stderr, "[dpgo_amd] ERROR: dpgo_group_create: node %d is not in [0, %d).\n", id, g->g.num_nodes);
      return -1;
    }
  return guarded([&] {
    auto *h = new dpgo_group();
    h->grp = new dpgo::Group(g->g, ids, to_cpp(*opt), device);
    if (!h->grp->ok()) {
      delete h->grp;
      delete h;
      return -1;
    }
    h->all.resize(num_local);
    std::iota(h->all.begin(), h->all.end(), 0);
    *out = h;
    return 0;
  });
}

void dpgo_group_free(dpgo_group_t *h) {
  if (!h) return;
  delete h->grp;
  delete h;
}

// the nodes a batched call works on: every node (locals == NULL), or a list of distinct local indices.
// false: an index is out of range or appears twice (res_[a] and the 64-bit node mask are indexed by it)
static bool sel(const dpgo_group_t *h, const int *locals, int n, std::vector<int> &out) {
  if (!locals || n <= 0) { out = h->all; return true; }
  out.assign(locals, locals + n);
  std::vector<char> seen(h->all.size(), 0);
  for (int a : out) {
    if (a < 0 || a >= (int)h->all.size() || seen[a]) {
      fprintf(stderr, "[dpgo_amd] ERROR: local node index %d is out of range [0, %zu) or repeated.\n", a, h->all.size());
      return false;
    }
    seen[a] = 1;
  }
  return true;
}

int dpgo_group_initialize(dpgo_group_t *h, int local, const double *X, int ld) { return guarded([&] { return h->grp->initialize(local, X, ld); }); }
int dpgo_group_initialize_global(dpgo_group_t *h, const double *X, int ld) { return guarded([&] { return h->grp->initialize_global(X, ld); }); }
int dpgo_group_update(dpgo_group_t *h, const int *locals, int n) {
  std::vector<int> v;
  if (!h || !sel(h, locals, n, v)) return -1;
  return guarded([&] { return h->grp->update(v); });
}
int dpgo_group_iterate(dpgo_group_t *h, const int *locals, int n) {
  std::vector<int> v;
  if (!h || !sel(h, locals, n, v)) return -1;
  return guarded([&] { return h->grp->iterate(v); });
}
int dpgo_group_communicate_local(dpgo_group_t *h) { return guarded([&] { return h->grp->communicate_local(); }); }
int dpgo_group_step(dpgo_group_t *h, struct dpgo_comm *comm) {
  if (!h) return -1;
  return guarded([&] {
    std::vector<int> all(h->grp->num_local());
    for (int a = 0; a < (int)all.size(); a++) all[a] = a;
    std::function<int()> xchg;
    if (comm && comm->c) xchg = [comm] { return comm->c->exchange(); };
    return h->grp->step(all, xchg);
  });
}
int dpgo_group_set_collectives(dpgo_group_t *h, void *send_dev, void *gathered_dev, dpgo_allgather_fn ag, dpgo_allreduce_fn ar,
                               void *user) {
  return guarded([&] { return h->grp->set_collectives((double *)send_dev, (double *)gathered_dev, ag, ar, user); });
}

int dpgo_group_star_initialize(dpgo_group_t *h, const double *X, int ld) { return guarded([&] { return h->grp->star_initialize_global(X, ld); }); }
int dpgo_group_star_update(dpgo_group_t *h) { return guarded([&] { return h->grp->star_update(); }); }
int dpgo_group_star_iterate(dpgo_group_t *h) { return guarded([&] { return h->grp->star_iterate(); }); }
int dpgo_group_star_state(const dpgo_group_t *h, double *F, double *fobj, double *fobjh, int *branches) {
  if (!h) return -1;
  return guarded([&] {
    if (F) *F = h->grp->star_F();
    if (fobj) *fobj = h->grp->star_fobj();
    if (fobjh) *fobjh = h->grp->star_fobjh();
    if (branches) *branches = h->grp->star_branches();
    return 0;
  });
}
int dpgo_group_receive(dpgo_group_t *h, int local, int beta, const double *msg, int ld) { return guarded([&] { return h->grp->receive(local, beta, msg, ld); }); }
int dpgo_group_send(const dpgo_group_t *h, int local, int beta, double *msg, int ld) { return guarded([&] { return h->grp->send(local, beta, msg, ld); }); }
int dpgo_group_message_sizes(const dpgo_group_t *h, int local, int beta, int *num_send, int *num_recv) {
  if (!h) return -1;
  return guarded([&] {
    const int s = h->grp->num_send(local, beta), r = h->grp->num_recv(local, beta);
    if (s < 0 || r < 0) return -1;
    if (num_send) *num_send = s;
    if (num_recv) *num_recv = r;
    return 0;
  });
}
int dpgo_group_num_sent(const dpgo_group_t *h) { return h->grp->num_sent(); }
int dpgo_group_sent_keys(const dpgo_group_t *h, int *nodes, int *poses) {
  const auto &k = h->grp->sent_keys();
  for (size_t i = 0; i < k.size(); i++) { nodes[i] = k[i].first; poses[i] = k[i].second; }
  return 0;
}
int dpgo_group_set_recv_layout(dpgo_group_t *h, int nranks, int stride, const int *counts, const int *nodes,
                               const int *poses) {
  return guarded([&] { return h->grp->set_recv_layout(nranks, stride, counts, nodes, poses); });
}
int dpgo_group_pack_sent(dpgo_group_t *h, void *buf) { return guarded([&] { return h->grp->pack_sent((double *)buf); }); }
int dpgo_group_unpack_recv(dpgo_group_t *h, const void *buf) { return guarded([&] { return h->grp->unpack_recv((const double *)buf); }); }
int dpgo_group_get_Xk(const dpgo_group_t *h, int local, double *X, int ld) { return guarded([&] { return h->grp->get_Xk(local, X, ld); }); }
int dpgo_group_get_Xak(const dpgo_group_t *h, int local, double *X, int ld) { return guarded([&] { return h->grp->get_X_own(local, X, ld); }); }
int dpgo_group_scatter_global(const dpgo_group_t *h, double *X, int ld) { return guarded([&] { return h->grp->scatter_global(X, ld); }); }
int dpgo_group_node_id(const dpgo_group_t *h, int local) {
  if (!h || local < 0 || local >= h->grp->num_local()) return -1;
  return h->grp->node_id(local);
}
int dpgo_group_sync(const dpgo_group_t *h) { return guarded([&] { h->grp->sync(); return 0; }); }
void *dpgo_group_stream(const dpgo_group_t *h) { return (void *)h->grp->stream(); }

int dpgo_group_results(const dpgo_group_t *h, int local, dpgo_results_t *o) {
  if (!h || !o || local < 0 || local >= h->grp->num_local()) return -1;
  // results() may have to take the deferred read-back of the last update(): a device error there becomes -1
  return guarded([&] {
  const dpgo::NodeResults &r = h->grp->results(local);
  o->updated = r.updated; o->iters = r.iters; o->gradFnorm = r.gradFnorm; o->fobjE = r.fobjE;
  o->Fk[0] = r.Fk[0]; o->Fk[1] = r.Fk[1]; o->Gk = r.Gk; o->Gkh = r.Gkh; o->fobj = r.fobj; o->f = r.f;
  o->gamma = r.gamma; o->s[0] = r.s0; o->s[1] = r.s1;
  o->soft_restart_hits[0] = r.soft_restart_hits[0]; o->soft_restart_hits[1] = r.soft_restart_hits[1];
  o->num_oscillations = r.num_oscillations; o->refined = r.refined; o->tnt_status = r.tnt_status;
  o->tnt_inner_iterations = r.tnt_inner; o->restarts = r.restarts;
  return 0;
  });
}

// ---- test hooks ----
static int ref_row(int d, int n0, int n1, int p, int slot) {
  if (p < n0) return slot == 0 ? p : n0 + p * d + slot - 1;
  const int k = p - n0;
  return slot == 0 ? (d + 1) * n0 + k : (d + 1) * n0 + n1 + k * d + slot - 1;
}

int dpgo_debug_node_matrix(const dpgo_graph_t *g, int node, const dpgo_options_t *opt, const char *name, int *rows,
                           int *cols, double *vals) {
  dpgo::DataInfo info;
  if (node_info(g, node, info) != 0) return -1;
  dpgo::NodeOperators ops;
  const bool trivial = opt->loss == 0;
  if (dpgo::assemble_node(info, opt->regularizer, trivial, ops) != 0) return -1;
  const int d = info.d, B = d + 1, n0 = info.n[0], n1 = info.n[1];
  const std::string nm(name);
  int cnt = 0;
  auto emit = [&](int p, int q, const double *blk) {
    for (int r = 0; r < B; r++)
      for (int c = 0; c < B; c++) {
        if (blk[r * B + c] == 0.0) continue;
        if (rows) { rows[cnt] = ref_row(d, n0, n1, p, r); cols[cnt] = ref_row(d, n0, n1, q, c); vals[cnt] = blk[r * B + c]; }
        cnt++;
      }
  };
  if (nm == "D") {
    for (int p = 0; p < n0; p++) emit(p, p, &ops.D[(size_t)p * B * B]);
    return cnt;
  }
  const dpgo::BsrMatrix *M = nm == "G" ? &ops.G : nm == "S" ? &ops.S : nm == "P" ? &ops.P : nm == "P0" ? &ops.P0
                                                                                    : nm == "Q" ? &ops.Q : nullptr;
  if (!M || M->B == 0) return -1;
  for (int p = 0; p < M->nrows; p++)
    for (int k = M->ptr[p]; k < M->ptr[p + 1]; k++) emit(p, M->col[k], &M->val[(size_t)k * B * B]);
  return cnt;
}

int dpgo_debug_node_proximal(const dpgo_graph_t *g, int node, const dpgo_options_t *opt, double *T, double *N,
                             double *V) {
  dpgo::DataInfo info;
  if (node_info(g, node, info) != 0) return -1;
  dpgo::NodeOperators ops;
  if (dpgo::assemble_node(info, opt->regularizer, opt->loss == 0, ops) != 0) return -1;
  std::copy(ops.Tinv.begin(), ops.Tinv.end(), T);
  std::copy(ops.N.begin(), ops.N.end(), N);
  std::copy(ops.V.begin(), ops.V.end(), V);
  return 0;
}

int dpgo_debug_spd_solve(int n, const int *ptr, const int *col, const double *val, double *X, int ncols, int leaf) {
  dpgo::CsrMatrix A;
  A.n = n;
  A.ptr.assign(ptr, ptr + n + 1);
  A.col.assign(col, col + ptr[n]);
  A.val.assign(val, val + ptr[n]);
  dpgo::SpdFactor F;
  if (dpgo::spd_factor(A, F, leaf) != 0) return -1;
  dpgo::spd_solve_host(F, X, ncols);
  return 0;
}

int dpgo_prof_enable(int on) { dpgo::prof_enable(on != 0); if (on) dpgo::prof_reset(); return 0; }
int dpgo_prof_num_kinds(void) { return dpgo::PK_COUNT; }
const char *dpgo_prof_kind_name(int k) {
  static const char *names[] = {"k_bsr", "k_inter", "k_proximal", "k_axpby", "k_dot", "k_rot_op", "k_copy_indexed",
                                "k_bdiag_dot", "k_reduce", "k_spd_fwd", "k_spd_bwd", "k_bsr_tcol"};
  static_assert(sizeof(names) / sizeof(names[0]) == dpgo::PK_COUNT, "one name per profiled kernel family");
  return (k >= 0 && k < dpgo::PK_COUNT) ? names[k] : "";
}
int dpgo_prof_collect(double *ms, double *bytes, long *count) { dpgo::prof_collect(ms, bytes, count); return 0; }
int dpgo_prof_collect_operands(double *operand_bytes) { if (!operand_bytes) return -1; dpgo::prof_collect_operands(operand_bytes); return 0; }
int dpgo_group_solver_stats(const dpgo_group_t *h, long *nnz_tt, long *nnz_rr, int *levels_tt, int *levels_rr) {
  *nnz_tt = (long)h->grp->factor_tt().nnz();
  *nnz_rr = (long)h->grp->factor_rr().nnz();
  *levels_tt = (int)h->grp->factor_tt().by_height.size();
  *levels_rr = (int)h->grp->factor_rr().by_height.size();
  return 0;
}

int dpgo_group_graph_stats(const dpgo_group_t *h, long *replays, long *captures, long *eager) {
  if (!h || !h->grp || !replays || !captures || !eager) return -1;
  h->grp->graph_stats(replays, captures, eager);
  return 0;
}

int dpgo_debug_spd_stats(int n, const int *ptr, const int *col, const double *val, int leaf, long *nnz, int *levels,
                         int *max_front) {
  dpgo::CsrMatrix A;
  A.n = n;
  A.ptr.assign(ptr, ptr + n + 1);
  A.col.assign(col, col + ptr[n]);
  A.val.assign(val, val + ptr[n]);
  dpgo::SpdFactor F;
  if (dpgo::spd_factor(A, F, leaf) != 0) return -1;
  if (const char *path = getenv("DPGO_SPD_DUMP_FRONTS")) {   // analysis hook: w u height depth per front
    if (FILE *fp = fopen(path, "w")) {
      for (int f = 0; f < F.nfronts; f++) fprintf(fp, "%d %d %d %d\n", F.w[f], F.u[f], F.height[f], F.depth[f]);
      fclose(fp);
    }
  }
  *nnz = (long)F.nnz();
  *levels = (int)F.by_height.size();
  *max_front = F.max_front;
  return 0;
}

int dpgo_group_debug_apply(dpgo_group_t *h, int local, const char *op, const double *in, int ld_in, double *out,
                           int ld_out) {
  return guarded([&] { return h->grp->debug_apply(local, op, in, ld_in, out, ld_out); });
}

// ---- boundary: DPGOStar::evaluate_f / evaluate_grad, set_options / options, problem() accessors, g2o export ----
int dpgo_group_evaluate(dpgo_group_t *h, const double *X, int ld, double *F, double *grad_sqnorm, double *grad, int ldg) {
  if (!h || !X) return -1;
  return guarded([&] { return h->grp->evaluate_global(X, ld, F, grad_sqnorm, grad, ldg); });
}

int dpgo_group_set_options(dpgo_group_t *h, const dpgo_options_t *opt) {
  if (!h || !opt) return -1;
  return guarded([&] { return h->grp->set_options(to_cpp(*opt)); });
}

int dpgo_group_get_options(const dpgo_group_t *h, dpgo_options_t *o) {
  if (!h || !o) return -1;
  const dpgo::Options &d = h->grp->options();
  o->scheme = d.scheme; o->regularizer = d.regularizer; o->accepted_delta = d.accepted_delta;
  o->eta[0] = d.eta[0]; o->eta[1] = d.eta[1]; o->psi = d.psi; o->phi = d.phi;
  o->max_soft_restart_hits[0] = d.max_soft_restart_hits[0]; o->max_soft_restart_hits[1] = d.max_soft_restart_hits[1];
  o->oscillation_cnt_period = d.oscillation_cnt_period; o->max_oscillations = d.max_oscillations;
  o->loss = d.loss; o->loss_reg = d.loss_reg; o->rescale = d.rescale; o->max_rescale_count = d.max_rescale_count;
  o->grad_norm_tol = d.grad_norm_tol;
  o->rel_func_decrease_tol = d.rel_func_decrease_tol; o->stepsize_tol = d.stepsize_tol;
  o->max_iterations = d.max_iterations; o->max_iterations_accepted = d.max_iterations_accepted;
  o->reg_Cholesky_precon_max_condition_number = d.reg_Cholesky_precon_max_condition_number;
  o->preconditioned_grad_norm_tol = d.preconditioned_grad_norm_tol; o->max_tCG_iterations = d.max_tCG_iterations;
  o->STPCG_kappa = d.STPCG_kappa; o->STPCG_theta = d.STPCG_theta; o->preconditioner = d.preconditioner;
  o->verbose = d.verbose;
  return 0;
}

int dpgo_graph_node_maps(const dpgo_graph_t *g, int node, int which, int *nodes, int *poses, int *block, int *local,
                         int *count) {
  dpgo::DataInfo info;
  if (!count || node_info(g, node, info) != 0 || which < 0 || which > 2) return -1;
  int k = 0;
  auto emit = [&](int b, int p, int blk, int loc) {
    if (nodes) { nodes[k] = b; poses[k] = p; block[k] = blk; local[k] = loc; }
    k++;
  };
  if (which == 0) {          // index_: own poses {0, k}, neighbour poses {1, k}   (DPGO_utils.cpp:400-418)
    for (int i = 0; i < info.n[0]; i++) emit(node, info.own_pose[i], 0, i);
    for (int i = 0; i < info.n[1]; i++) emit(info.nbr_key[i].first, info.nbr_key[i].second, 1, i);
  } else if (which == 1) {   // sent_[beta][own pose] = {0, k}                     (DPGO_utils.cpp:428-431)
    for (const auto &s : info.sent)
      for (int i : s.second) emit(s.first, info.own_pose[i], 0, i);
  } else {                   // recv_[beta][pose of beta] = {1, k}                 (DPGO_utils.cpp:432-435)
    for (const auto &r : info.recv)
      for (const auto &pk : r.second) emit(r.first, pk.first, 1, pk.second);
  }
  *count = k;
  return 0;
}

// g2o export: VERTEX_SE2 / VERTEX_SE3:QUAT lines from X plus the graph's edges.  The information matrices are the
// isotropic ones the loader's formulas invert (tau = d / tr(I_t^-1), kappa = I33 | 3 / (2 tr(I_R^-1)),
// DPGO_utils.cpp:63-67, 107-116), so reading the file back gives the same (R, t, kappa, tau).
int dpgo_write_g2o(const dpgo_graph_t *g, const double *X, int ld, const char *filename) {
  if (!g || !filename) return -1;
  const int d = g->g.d, N = g->g.num_poses;
  if (X && ld < (d + 1) * N) return -1;
  FILE *fp = fopen(filename, "w");
  if (!fp) {
    fprintf(stderr, "[dpgo_amd] ERROR: cannot open %s for writing.\n", filename);
    return -1;
  }
  auto quat = [](const double *R, double *q) {   // R row-major 3x3 -> (qx, qy, qz, qw)
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
      const double s = std::sqrt(tr + 1.0) * 2;
      q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
      const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
      q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
      const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
      q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s;
    } else {
      const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
      q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s;
    }
  };
  if (X)
    for (int i = 0; i < N; i++) {
      // rows N + d i .. of X hold R_i^T: R_i(r, c) = X(N + d i + c, r)
      double R[9], t[3];
      for (int c = 0; c < d; c++) {
        t[c] = X[(size_t)c * ld + i];
        for (int r = 0; r < d; r++) R[r * d + c] = X[(size_t)r * ld + N + i * d + c];
      }
      if (d == 2) fprintf(fp, "VERTEX_SE2 %d %.17g %.17g %.17g\n", i, t[0], t[1], std::atan2(R[2], R[0]));
      else {
        double q[4];
        quat(R, q);
        fprintf(fp, "VERTEX_SE3:QUAT %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", i, t[0], t[1], t[2], q[0], q[1], q[2], q[3]);
      }
    }
  for (const auto &m : g->g.all) {
    if (d == 2) {
      fprintf(fp, "EDGE_SE2 %d %d %.17g %.17g %.17g %.17g 0 0 %.17g 0 %.17g\n", m.ipose, m.jpose, m.t[0], m.t[1],
              std::atan2(m.R[2], m.R[0]), m.tau, m.tau, m.kappa);
    } else {
      double q[4];
      quat(m.R, q);
      fprintf(fp, "EDGE_SE3:QUAT %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g", m.ipose, m.jpose, m.t[0], m.t[1], m.t[2], q[0], q[1],
              q[2], q[3]);
      for (int r = 0; r < 6; r++)
        for (int c = r; c < 6; c++) fprintf(fp, " %.17g", r == c ? (r < 3 ? m.tau : 2.0 * m.kappa) : 0.0);
      fprintf(fp, "\n");
    }
  }
  fclose(fp);
  return 0;
}

// test hook: the neighbour-to-neighbour exchange plan of rank `rank` (comm.cpp: p2p_plan) from every rank's exported and
// needed keys.  exp_counts / need_counts: keys per rank; *_nodes / *_poses: the keys, rank after rank.
// Out (may be null to query sizes): npeers; per peer 5 ints (rank, send_off, send_cnt, recv_off, recv_cnt); the send and
// the receive keys (node, pose interleaved).  sizes[0..2] = npeers, send keys, recv keys.
int dpgo_debug_p2p_plan(int rank, int nranks, const int *exp_counts, const int *exp_nodes, const int *exp_poses,
                        const int *need_counts, const int *need_nodes, const int *need_poses, int *peers, int *send_keys,
                        int *recv_keys, int *sizes) {
  if (nranks < 1 || rank < 0 || rank >= nranks || !exp_counts || !need_counts || !sizes) return -1;
  return guarded([&] {
    std::vector<std::vector<dpgo::PoseKey>> ex(nranks), nd(nranks);
    size_t a = 0, b = 0;
    for (int r = 0; r < nranks; r++) {
      for (int k = 0; k < exp_counts[r]; k++, a++) ex[r].push_back({exp_nodes[a], exp_poses[a]});
      for (int k = 0; k < need_counts[r]; k++, b++) nd[r].push_back({need_nodes[b], need_poses[b]});
    }
    const dpgo::P2PPlan P = dpgo::p2p_plan(rank, ex, nd);
    sizes[0] = (int)P.peers.size(); sizes[1] = (int)P.send_keys.size(); sizes[2] = (int)P.recv_keys.size();
    if (peers)
      for (size_t i = 0; i < P.peers.size(); i++) {
        const auto &q = P.peers[i];
        const int v[5] = {q.rank, q.send_off, q.send_cnt, q.recv_off, q.recv_cnt};
        std::copy(v, v + 5, peers + 5 * i);
      }
    if (send_keys) for (size_t i = 0; i < P.send_keys.size(); i++) { send_keys[2 * i] = P.send_keys[i].first; send_keys[2 * i + 1] = P.send_keys[i].second; }
    if (recv_keys) for (size_t i = 0; i < P.recv_keys.size(); i++) { recv_keys[2 * i] = P.recv_keys[i].first; recv_keys[2 * i + 1] = P.recv_keys[i].second; }
    return 0;
  });
}

// ---- RCCL exchange (comm.cpp) ----
int dpgo_comm_unique_id(void *id128) {
  if (!id128) return -1;
  return guarded([&] { return dpgo::Comm::unique_id(id128); });
}

int dpgo_comm_create(dpgo_group_t *h, int rank, int nranks, const void *id128, dpgo_comm_t **out) {
  if (!out) return -1;
  *out = nullptr;
  if (!h || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return -1;
  return guarded([&] {
    std::unique_ptr<dpgo_comm> c(new dpgo_comm());
    c->c = nullptr;
    std::unique_ptr<dpgo::Comm> cc(new dpgo::Comm(h->grp, rank, nranks, id128));   // (cleans up after itself when it throws)
    if (!cc->ok()) return -1;
    c->c = cc.release();
    *out = c.release();
    return 0;
  });
}

void dpgo_comm_free(dpgo_comm_t *c) {
  if (!c) return;
  (void)guarded([&] { delete c->c; return 0; });
  delete c;
}

int dpgo_comm_exchange(dpgo_comm_t *c) {
  if (!c) return -1;
  return guarded([&] { return c->c->exchange(); });
}

int dpgo_comm_allreduce_sum(dpgo_comm_t *c, double *vals, long n) {
  if (!c || !vals || n < 0) return -1;
  return guarded([&] { return n <= 4096 ? c->c->allreduce(vals, (int)n) : c->c->allreduce_large(vals, (size_t)n); });
}

int dpgo_comm_exchange_kind(const dpgo_comm_t *c) {
  if (!c || !c->c) return -1;
  return std::string(c->c->exchange_kind()) == "p2p" ? 1 : 0;
}

long dpgo_comm_bytes_sent(const dpgo_comm_t *c) {
  if (!c || !c->c) return -1;
  return (long)c->c->bytes_sent_per_exchange();
}

int dpgo_debug_comm_p2p_self(dpgo_group_t *h) {
  if (!h) return -1;
  return guarded([&] {
    unsigned char id[128];
    if (dpgo::Comm::unique_id(id) != 0) return -1;
    dpgo::Comm c(h->grp, 0, 1, id, /*layout=*/false);   // a communicator of one rank: the group's neighbours need no host
    return c.p2p_self_check();
  });
}

int dpgo_comm_create_self(dpgo_group_t *h, dpgo_comm_t **out) {
  if (!out) return -1;
  *out = nullptr;
  if (!h) return -1;
  return guarded([&] {
    unsigned char id[128];
    if (dpgo::Comm::unique_id(id) != 0) return -1;
    std::unique_ptr<dpgo_comm> c(new dpgo_comm());
    std::unique_ptr<dpgo::Comm> cc(new dpgo::Comm(h->grp, 0, 1, id, /*layout=*/false));
    if (cc->enable_self_exchange() != 0) return -1;
    c->c = cc.release();
    *out = c.release();
    return 0;
  });
}
int dpgo_comm_self_exchange(dpgo_comm_t *c) {
  if (!c || !c->c) return -1;
  return guarded([&] { return c->c->enable_self_exchange(); });
}
int dpgo_comm_enable_timing(dpgo_comm_t *c) {
  if (!c || !c->c) return -1;
  return guarded([&] { return c->c->enable_timing(); });
}
int dpgo_comm_exchange_time(dpgo_comm_t *c, double *mean_us, long *count) {
  if (!c || !c->c) return -1;
  return guarded([&] { return c->c->exchange_time(mean_us, count); });
}

int dpgo_comm_barrier(dpgo_comm_t *c) {
  if (!c) return -1;
  return guarded([&] { return c->c->barrier(); });
}

int dpgo_host_pack_sent(const dpgo_graph_t *g, const int *node_ids, int num_local, const double *X, int ld, double *buf) {
  if (!g || !node_ids || num_local <= 0 || !X || !buf) return -1;
  const int d = g->g.d, N = g->g.num_poses, RS = (d + 1) * d;
  if (ld < (d + 1) * N) return -1;
  int cnt[2];
  if (dpgo_graph_exchange_plan(g, node_ids, num_local, nullptr, nullptr, nullptr, nullptr, cnt) != 0) return -1;
  std::vector<int> sn(cnt[0]), sp(cnt[0]), rn(cnt[1]), rp(cnt[1]);
  if (dpgo_graph_exchange_plan(g, node_ids, num_local, sn.data(), sp.data(), rn.data(), rp.data(), cnt) != 0) return -1;
  for (int k = 0; k < cnt[0]; k++) {
    const int gid = g->g.g_index[sn[k]].at(sp[k]);
    for (int c = 0; c < d; c++) {
      buf[(size_t)k * RS + c] = X[(size_t)c * ld + gid];
      for (int r = 0; r < d; r++) buf[(size_t)k * RS + d + r * d + c] = X[(size_t)c * ld + N + gid * d + r];
    }
  }
  return cnt[0];
}

int dpgo_host_unpack_recv(const dpgo_graph_t *g, const int *node_ids, int num_local, int node, int nranks, int stride,
                          const int *counts, const int *nodes, const int *poses, const double *gathered, double *Z, int ldz) {
  if (!g || !node_ids || !counts || !nodes || !poses || !gathered || !Z || nranks < 1 || stride < 1) return -1;
  dpgo::DataInfo info;
  if (node_info(g, node, info) != 0) return -1;
  const int d = g->g.d, RS = (d + 1) * d, n0 = info.n[0], n1 = info.n[1];
  if (ldz < (d + 1) * (n0 + n1)) return -1;
  std::set<int> local(node_ids, node_ids + num_local);
  std::map<std::pair<int, int>, int> slot;
  int off = 0;
  for (int r = 0; r < nranks; r++) {
    for (int k = 0; k < counts[r]; k++) slot[{nodes[off + k], poses[off + k]}] = r * stride + k;
    off += counts[r];
  }
  int written = 0;
  for (int k = 0; k < n1; k++) {
    const auto key = info.nbr_key[k];
    if (local.count(key.first)) continue;
    auto it = slot.find(key);
    if (it == slot.end()) {
      fprintf(stderr, "[dpgo_amd] ERROR: No information for pose [%d, %d].\n", key.first, key.second);
      return -1;
    }
    const double *rec = gathered + (size_t)it->second * RS;
    for (int c = 0; c < d; c++) {
      Z[(size_t)c * ldz + (d + 1) * n0 + k] = rec[c];
      for (int r = 0; r < d; r++) Z[(size_t)c * ldz + (d + 1) * n0 + n1 + k * d + r] = rec[d + r * d + c];
    }
    written++;
  }
  return written;
}

void dpgo_dchordal_options_default(dpgo_dchordal_options_t *o) {
  const dpgo::DChordalOptions d;
  for (int k = 0; k < 4; k++) o->iters[k] = d.iters[k];
  o->local_iters = d.local_iters;
  o->reg_G = d.reg_G;
}

int dpgo_group_dist_chordal_initialization(dpgo_group_t *h, const dpgo_dchordal_options_t *opt, const double *X_local,
                                           int ld_local, double *X, int ld, double *objectives, int *num_objectives) {
  if (!h || !X) return -1;
  dpgo::DChordalOptions o;
  if (opt) {
    for (int k = 0; k < 4; k++) o.iters[k] = opt->iters[k];
    o.local_iters = opt->local_iters;
    o.reg_G = opt->reg_G;
    if (o.reg_G < 0 || o.local_iters < 0 || o.iters[0] < 0 || o.iters[1] < 0 || o.iters[2] < 0 || o.iters[3] < 0) return -1;
  }
  return guarded([&] {
    std::vector<double> obj;
    const int rc = h->grp->dist_chordal_initialization(o, X_local, ld_local, X, ld, objectives ? &obj : nullptr);
    if (rc == 0 && objectives && num_objectives) {
      const int n = std::min<int>(*num_objectives, (int)obj.size());
      std::copy(obj.begin(), obj.begin() + n, objectives);
      *num_objectives = (int)obj.size();
    }
    return rc;
  });
}

}  // extern "C"
