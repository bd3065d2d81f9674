// dpgo_amd.hpp -- header-only C++ facade over the C ABI (dpgo_amd.h) with the reference's names.
//
// What a C++ caller of MurpheyLab/DPGO sees, minus Eigen: DPGO::Matrix is a small column-major
// owner of doubles (the layout Eigen::MatrixXd has by default), everything else keeps the
// reference's spelling and meaning:
//
//   reference (C++/DPGO/include/DPGO)                  here (namespace DPGO)
//   -------------------------------------------------  ----------------------------------------------
//   read_g2o(...)                  DPGO_utils.h:49-51   Graph::read_g2o(filename, num_nodes)
//   Options                        DPGO_types.h:78-201  Options (same field names, same defaults)
//   DPGOResult (scalars, Xk, Xak)  DPGO_types.h:204-322 DPGOResult
//   DPGOHash(node, meas, options)  DPGOHash.h:13-107    DPGOHash  = one node of a DPGOHashGroup
//     initialize / update / iterate / communicate / receive / results / options
//   vector<shared_ptr<DPGOHash>>   dist_pgo.cpp:96      DPGOHashGroup (the nodes one GPU hosts; batched calls)
//   DPGOStar                       DPGOStar.h:13-61     DPGOStar (AMM-PGO*, all nodes in one group)
//
// Return codes follow the reference: 0 ok, -1 error (a line on stderr).  Constructors throw
// std::runtime_error (there is no CPU fallback: no HIP device => no group).
#ifndef DPGO_AMD_HPP
#define DPGO_AMD_HPP

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "dpgo_amd.h"

namespace DPGO {

using Scalar = double;

// column-major dense matrix (Eigen::MatrixXd layout)
class Matrix {
 public:
  Matrix() = default;
  Matrix(int rows, int cols) : rows_(rows), cols_(cols), v_((size_t)rows * cols, 0.0) {}
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  Scalar *data() { return v_.data(); }
  const Scalar *data() const { return v_.data(); }
  Scalar &operator()(int r, int c) { return v_[(size_t)c * rows_ + r]; }
  Scalar operator()(int r, int c) const { return v_[(size_t)c * rows_ + r]; }
  void resize(int rows, int cols) { rows_ = rows; cols_ = cols; v_.assign((size_t)rows * cols, 0.0); }

 private:
  int rows_ = 0, cols_ = 0;
  std::vector<Scalar> v_;
};

enum class Scheme { MM = 0, AMM = 1 };                                       // DPGO_types.h:52
enum class Loss { None = 0, Huber = 1, GemanMcClure = 2, Welsch = 3 };       // DPGO_types.h:54
enum class Preconditioner { None = 0, Jacobi = 1, IncompleteCholesky = 2, RegularizedCholesky = 3 };   // DPGO_types.h:35-40
enum class Rescale { Static = 0, Dynamic = 1 };                              // DPGO_types.h:43-46

// entry of DPGOProblem::index() / sent() / recv() (DPGOProblem.h:212-225): (node, pose) -> {block, k}
struct IndexEntry {
  int node, pose, block, k;
};

// DPGO::Options (DPGO_types.h:78-201): the C struct with the reference's field names
struct Options : dpgo_options_t {
  Options() { dpgo_options_default(this); }
  // the overrides of C++/examples/dist_pgo.cpp:103-120
  static Options driver(Loss loss = Loss::None, bool accelerated = true) {
    Options o;
    dpgo_options_driver(&o, (int)loss, accelerated ? 1 : 0);
    return o;
  }
};

// the partitioned measurements of read_g2o + generate_data_info (host only)
class Graph {
 public:
  static std::shared_ptr<Graph> read_g2o(const std::string &filename, int num_nodes) {
    dpgo_graph_t *h = nullptr;
    if (dpgo_read_g2o(filename.c_str(), num_nodes, &h) != 0) throw std::runtime_error("DPGO::read_g2o: " + filename);
    return std::shared_ptr<Graph>(new Graph(h));
  }
  ~Graph() { dpgo_graph_free(h_); }
  Graph(const Graph &) = delete;
  Graph &operator=(const Graph &) = delete;
  int d() const { return d_; }
  int num_poses() const { return num_poses_; }
  int num_nodes() const { return num_nodes_; }
  int num_edges() const { return num_edges_; }
  // DPGOProblem::n() / m() (DPGOProblem.h:241-248): {own, neighbour} poses, {intra, inter} edges
  void sizes(int node, int n[2], int m[2]) const { dpgo_graph_node_sizes(h_, node, &n[0], &n[1], &m[0], &m[1]); }
  int offset(int node) const { return dpgo_graph_node_offset(h_, node); }   // first global pose id of the node
  // DPGOProblem::index() / sent() / recv() of a node (DPGOProblem.h:212-225), entries in map order
  std::vector<IndexEntry> index(int node) const { return maps(node, 0); }
  std::vector<IndexEntry> sent(int node) const { return maps(node, 1); }
  std::vector<IndexEntry> recv(int node) const { return maps(node, 2); }
  // g2o export: VERTEX_* lines from X ((d+1) N x d) + the EDGE_* lines
  int write_g2o(const std::string &filename, const Matrix &X) const {
    return dpgo_write_g2o(h_, X.data(), X.rows(), filename.c_str());
  }
  // centralised chordal initialisation (dist_pgo.cpp:416-444): X is (d+1) N x d
  Matrix chordal_initialization() const {
    Matrix X((d_ + 1) * num_poses_, d_);
    if (dpgo_chordal_initialization(h_, X.data(), X.rows()) != 0) throw std::runtime_error("chordal_initialization");
    return X;
  }
  const dpgo_graph_t *handle() const { return h_; }

 private:
  explicit Graph(dpgo_graph_t *h) : h_(h) { dpgo_graph_info(h_, &d_, &num_poses_, &num_nodes_, &num_edges_); }
  std::vector<IndexEntry> maps(int node, int which) const {
    int cnt = 0;
    if (dpgo_graph_node_maps(h_, node, which, nullptr, nullptr, nullptr, nullptr, &cnt) != 0) return {};
    std::vector<int> a(cnt), b(cnt), c(cnt), d(cnt);
    dpgo_graph_node_maps(h_, node, which, a.data(), b.data(), c.data(), d.data(), &cnt);
    std::vector<IndexEntry> out(cnt);
    for (int i = 0; i < cnt; i++) out[i] = {a[i], b[i], c[i], d[i]};
    return out;
  }
  dpgo_graph_t *h_;
  int d_ = 0, num_poses_ = 0, num_nodes_ = 0, num_edges_ = 0;
};

// scalar part of DPGOResult + on-demand copies of Xk / Xak (DPGO_types.h:204-322)
struct DPGOResult : dpgo_results_t {
  Matrix Xk, Xak;
};

class DPGOHashGroup;

// One node: the reference's DPGOHash interface (DPGOHash.h:13-107) on top of its group.
class DPGOHash {
 public:
  int node() const;
  int initialize(const Matrix &X) const;             // (d+1)(n0+n1) x d            DPGOHash.cpp:20-43
  int update() const;                                //                               :84-228
  int iterate() const;                               //                               :583-628
  // message from neighbour node beta: ((d+1) |recv[beta]|) x d, [t rows ; R rows]    :45-82
  int receive(int beta, const Matrix &msg) const;
  Matrix send(int beta) const;                       // what this node owes beta (sent[beta], DPGO_utils.cpp:428-435)
  DPGOResult results(bool with_X = true) const;      // results().Xk is what peers and the driver read
  const Options &options() const;

 private:
  friend class DPGOHashGroup;
  DPGOHash(DPGOHashGroup *g, int local) : g_(g), local_(local) {}
  DPGOHashGroup *g_;
  int local_;
};

// The nodes hosted by one GPU.  Batched update()/iterate()/communicate() replace the driver's loops over
// alpha (dist_pgo.cpp:455-462, 496-521); operator[] gives the per-node view.
class DPGOHashGroup {
 public:
  DPGOHashGroup(std::shared_ptr<Graph> graph, const std::vector<int> &nodes, const Options &options, int device = 0)
      : graph_(std::move(graph)), nodes_(nodes), options_(options) {
    if (dpgo_group_create(graph_->handle(), nodes_.data(), (int)nodes_.size(), &options_, device, &h_) != 0)
      throw std::runtime_error("dpgo_group_create failed (no HIP device, or inconsistent input); there is no CPU path");
    for (int k = 0; k < (int)nodes_.size(); k++) hash_.push_back(DPGOHash(this, k));
  }
  ~DPGOHashGroup() { dpgo_group_free(h_); }
  DPGOHashGroup(const DPGOHashGroup &) = delete;
  DPGOHashGroup &operator=(const DPGOHashGroup &) = delete;

  size_t size() const { return nodes_.size(); }
  const DPGOHash &operator[](int local) const { return hash_[local]; }
  // split a global X over the nodes and fill the neighbour rows (dist_pgo.cpp:435-446 + DPGO::communicate)
  int initialize(const Matrix &X) { return dpgo_group_initialize_global(h_, X.data(), X.rows()); }
  int update() { return dpgo_group_update(h_, nullptr, 0); }
  int iterate() { return dpgo_group_iterate(h_, nullptr, 0); }
  int communicate() { return dpgo_group_communicate_local(h_); }   // neighbours hosted by this group
  // gather X^alpha into the global X (dist_pgo.cpp:502-511)
  int gather(Matrix &X) const { return dpgo_group_scatter_global(h_, X.data(), X.rows()); }
  const Options &options() const { return options_; }
  // DPGOHash::set_options (DPGOHash.h:93-96) for every node of the group
  int set_options(const Options &o) {
    if (dpgo_group_set_options(h_, &o) != 0) return -1;
    options_ = o;
    return 0;
  }
  // DPGOStar::evaluate_f / evaluate_grad at an arbitrary global X (DPGOStar.cpp:713-829), summed over this group
  int evaluate_f(const Matrix &X, Scalar &fobj) const { return dpgo_group_evaluate(h_, X.data(), X.rows(), &fobj, nullptr, nullptr, 0); }
  int evaluate_grad(const Matrix &X, Matrix &grad) const {
    grad.resize(X.rows(), X.cols());
    return dpgo_group_evaluate(h_, X.data(), X.rows(), nullptr, nullptr, grad.data(), grad.rows());
  }
  const Graph &graph() const { return *graph_; }
  dpgo_group_t *handle() const { return h_; }

 private:
  friend class DPGOHash;
  std::shared_ptr<Graph> graph_;
  std::vector<int> nodes_;
  Options options_;
  dpgo_group_t *h_ = nullptr;
  std::vector<DPGOHash> hash_;
};

inline int DPGOHash::node() const { return dpgo_group_node_id(g_->h_, local_); }
inline int DPGOHash::initialize(const Matrix &X) const { return dpgo_group_initialize(g_->h_, local_, X.data(), X.rows()); }
inline int DPGOHash::update() const { return dpgo_group_update(g_->h_, &local_, 1); }
inline int DPGOHash::iterate() const { return dpgo_group_iterate(g_->h_, &local_, 1); }
inline int DPGOHash::receive(int beta, const Matrix &msg) const {
  return dpgo_group_receive(g_->h_, local_, beta, msg.data(), msg.rows());
}
inline Matrix DPGOHash::send(int beta) const {
  int ns = 0, nr = 0;
  if (dpgo_group_message_sizes(g_->h_, local_, beta, &ns, &nr) != 0) return Matrix();
  const int d = g_->graph_->d();
  Matrix msg((d + 1) * ns, d);
  if (ns > 0) dpgo_group_send(g_->h_, local_, beta, msg.data(), msg.rows());
  return msg;
}
inline DPGOResult DPGOHash::results(bool with_X) const {
  DPGOResult r;
  dpgo_group_results(g_->h_, local_, &r);
  if (with_X) {
    int n[2], m[2];
    g_->graph_->sizes(node(), n, m);
    const int d = g_->graph_->d(), rows = (d + 1) * (n[0] + n[1]);
    r.Xk.resize(rows, d);
    r.Xak.resize((d + 1) * n[0], d);
    dpgo_group_get_Xk(g_->h_, local_, r.Xk.data(), rows);
    dpgo_group_get_Xak(g_->h_, local_, r.Xak.data(), r.Xak.rows());
  }
  return r;
}
inline const Options &DPGOHash::options() const { return g_->options_; }

// AMM-PGO* (DPGOStar.h:13-61): initialize / update / iterate / communicate; every node lives in one group.
class DPGOStar {
 public:
  DPGOStar(int num_nodes, const std::string &filename, const Options &options, int device = 0)
      : graph_(Graph::read_g2o(filename, num_nodes)) {
    std::vector<int> all(num_nodes);
    for (int a = 0; a < num_nodes; a++) all[a] = a;
    group_.reset(new DPGOHashGroup(graph_, all, options, device));
  }
  int initialize(const Matrix &X) { return dpgo_group_star_initialize(group_->handle(), X.data(), X.rows()); }
  int update() { return dpgo_group_star_update(group_->handle()); }
  int iterate() { return dpgo_group_star_iterate(group_->handle()); }
  int communicate() { return group_->communicate(); }
  // F: running average (DPGOStar.cpp:210); fobj = F(X_k+1); fobjh = F(X_k+1/2)
  int state(double &F, double &fobj, double &fobjh, int &branches) const {
    return dpgo_group_star_state(group_->handle(), &F, &fobj, &fobjh, &branches);
  }
  // DPGOStar::evaluate_f / evaluate_grad (DPGOStar.h:49-51, DPGOStar.cpp:713-829): any X of size (d+1) N x d
  int evaluate_f(const Matrix &X, Scalar &fobj) const { return group_->evaluate_f(X, fobj); }
  int evaluate_grad(const Matrix &X, Matrix &grad) const { return group_->evaluate_grad(X, grad); }
  const Options &options() const { return group_->options(); }
  int set_options(const Options &o) { return group_->set_options(o); }
  int num_nodes() const { return graph_->num_nodes(); }
  const DPGOHashGroup &nodes() const { return *group_; }
  const Graph &graph() const { return *graph_; }

 private:
  std::shared_ptr<Graph> graph_;
  std::unique_ptr<DPGOHashGroup> group_;
};

}  // namespace DPGO

#endif  // DPGO_AMD_HPP
