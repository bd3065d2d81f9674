// A group of DPGO nodes hosted on one MI355X.
//
// Host-side mirror of the reference's per-node optimizer
//   DPGOHash    C++/DPGO/src/DPGOHash.cpp (initialize :20-43, update :84-228,
//               amm_pgo :230-444, mm_pgo :446-581, iterate :583-628),
//               C++/DPGO/include/DPGO/DPGOHash.h:28-86 (communicate)
//   DPGOProblem C++/DPGO/src/DPGOProblem.cpp (operators), DPGOProblem.h:275-294
// with every vector operation batched over the nodes of the group and executed
// by the kernels of kernels.hip.  The scalar state machine (Nesterov sequence,
// adaptive-restart counters) stays on the host exactly as in the reference; a
// per-node device mask lets nodes that take different branches share launches.
#pragma once
#include <chrono>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <stdexcept>
#include <utility>
#include <string>
#include <vector>

#include "assemble.h"
#include "graph.h"
#include "kernels.h"
#include "spd.h"

namespace dpgo {

// thrown by the host layer when a HIP call fails; the C ABI catches it and returns -1
struct DeviceError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// DPGO::Options (C++/DPGO/include/DPGO/DPGO_types.h:78-201), plain data.
struct Options {
  int scheme = 1;  // 0 MM, 1 AMM
  double regularizer = 1e-10;
  double accepted_delta = 5e-4;
  double eta[2] = {5e-4, 2.5e-2};
  double psi = 1e-10;
  double phi = 1e-6;
  int max_soft_restart_hits[2] = {10, 25};
  int oscillation_cnt_period = 15;
  int max_oscillations = 12;
  int loss = 0;  // 0 None, 1 Huber, 2 GemanMcClure, 3 Welsch
  double loss_reg = 1.0;
  int rescale = 1;            // 0 Static, 1 Dynamic (DPGO_types.h:128)
  int max_rescale_count = 5;  // DPGO_types.h:131
  double grad_norm_tol = 5e-3;
  double rel_func_decrease_tol = 1e-6;
  double stepsize_tol = 1e-4;
  int max_iterations = 10;
  int max_iterations_accepted = 1;
  double reg_Cholesky_precon_max_condition_number = 1e6;
  double preconditioned_grad_norm_tol = 1e-4;
  int max_tCG_iterations = 10000;
  double STPCG_kappa = 0.05;
  double STPCG_theta = 0.9;
  int verbose = 0;         // DPGO_types.h:87: a summary line per node and TNT refinement on stdout
  int preconditioner = 3;  // Preconditioner (DPGO_types.h:35-40): 0 None, 1 Jacobi, 2 IncompleteCholesky, 3 RegularizedCholesky
};

// The scalar fields of DPGOResult (DPGO_types.h:204-322) the state machine uses.
struct NodeResults {
  int updated = 1;
  int iters = 0;
  bool pre_done = false, pre_repeat = false;   // host_update_pre ran for the update under way (and what `repeat` was)
  int hist_iter = -1;   // iteration whose X[iter-1], g[iter-1], fobj[iter-1], s[iter] are in place (update() may run twice per iteration)
  double gradFnorm = 0, fobjE = 0, Fk[2] = {0, 0}, Gk = 0, Gkh = 0;
  double Gk_alt = 0;   // run_tnt: the refined point's surrogate value under the caller's second linear term
  double fobj = 0, fobj_prev = 0, f = 0, gamma = 0, s0 = 1, s1 = 1;
  int soft_restart_hits[2] = {0, 0};
  int num_oscillations = 0;
  int refined = 0;
  int tnt_status = -1;
  int tnt_inner = 0;
  int restarts = 0;
  std::vector<int> oscillations;
};

// DChordal::Options (C++/DChordal/include/DChordal/DChordal_types.h:44-70) + the driver's stage schedule
// (C++/examples/dist_pgo.cpp:205,274,344,383) + the length of the stage-0 stand-in (dchordal.cpp)
struct DChordalOptions {
  int iters[4] = {100, 400, 150, 250};   // reduced R, R, reduced t, t
  int local_iters = 30;
  double reg_G = 1e-12;
};

// After this, DevBuf never calls hipFree again in this process: a stuck RCCL kernel that cannot be aborted would make
// every hipFree wait for ever (comm.cpp: Comm::abandon).
void dev_leak_buffers(bool on);
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() {}
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release();
  void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
  void alloc(size_t count, bool zero = true);
  void upload(const std::vector<T> &h);
  void download(std::vector<T> &h) const;
};

// DPGO_SETUP_TIMING=1: wall time of the set-up phases on stderr
struct SetupClock {
  const bool on = getenv("DPGO_SETUP_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[setup] %-44s %8.3f s\n", what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

struct SpdSolverDev {
  SpdFactor F;   // host copy kept for sizes / host solves
  ~SpdSolverDev() { spd_release_device(F); spd_release_numeric(F); }
  DevBuf<int> piv_idx, upd_idx, asm_ptr, ubuf_dst;
  DevBuf<double> W, WT, ubuf, ytmp;   // W / WT: backward / forward panels (see upload)
  DevBuf<SpdItem> fwd_items, bwd_items;
  // one launch.  Tiles [tile0, tile0 + nwide + nnarrow) of the sweep's item list, stored node by node: node a's wide
  // tiles (rows high) are [wstart[a], wstart[a] + wcount[a]), its narrow ones [nstart[a], nstart[a] + ncount[a])
  struct Level {
    int tile0, nwide, nnarrow, rows;
    std::vector<int> wstart, wcount, nstart, ncount;
    std::vector<double> node_bytes;   // algorithmic bytes of the level per node
    // the launch for the nodes of `bits` (false: none of them has a tile here); bytes: their share of the level
    bool map(NodeBits bits, SpdLevelMap &M, double *bytes = nullptr) const;
  };
  std::vector<Level> fwd_levels, bwd_levels;   // (without the roots of the trees)
  // the roots: forward and backward step fused into one launch over the explicit inverse of the root's Schur
  // complement (k_spd_level MODE 2); DPGO_SPD_FUSE_ROOT=0 keeps them in the two sweeps
  Level root_level{0, 0, 0, 64, {}, {}, {}, {}, {}};
  bool fused_root = false;
  DevBuf<SpdItem> root_items;
  // The fused roots stored as ONE TRIANGLE of 64 x 64 blocks (kernels.h: RootRow; k_root_sym + k_root_combine): half the
  // bytes of the level for one small launch more, taken when a single root holds at least DPGO_SPD_ROOT_SYM_MB (32) megabytes
  // (DPGO_SPD_ROOT_SYM=1 / 0 forces it on / off).  root_items are then the wave-sized items, root_sym_level / root_rows_level
  // their and the block rows' per-node ranges, root_part the partial-sum slots, root_pack the per-block descriptors the
  // panels are cut with (also by repack()).
  // the next finer tile class of the full-product roots, for launches over few live roots (upload(), spd_run)
  Level root_fine_level{0, 0, 0, 16, {}, {}, {}, {}, {}};
  DevBuf<SpdItem> root_fine_items;
  DevBuf<double> Wroot_fine;
  int root_fine_rows = 0, root_fine_below = 0;
  bool fine_root_for(NodeBits v) const;
  bool root_sym = false;
  Level root_sym_level{0, 0, 0, 64, {}, {}, {}, {}, {}}, root_rows_level{0, 0, 0, 64, {}, {}, {}, {}, {}};
  DevBuf<RootRow> root_rows;
  DevBuf<double> root_part;
  DevBuf<SpdItem> root_pack;
  DevBuf<double> Wroot, Proot;   // the root tiles' panels; the dense products they are cut from (kept with keep_numeric)
  DevBuf<RootDesc> root_desc;
  int root_max_w = 0;
  std::vector<double> fwd_level_bytes, bwd_level_bytes;
  SpdDev dev;
  int dof = 1;
  bool stream_once = true;   // panels read with non-temporal loads (see upload)
  void upload(int dcols, const std::vector<int> &node_of_unknown);   // node_of_unknown: local node of every row of A
  DevBuf<PanelSrc> fwd_srcs, bwd_srcs, root_srcs;   // where every tile's panel comes from in the front-major factor
  int repack(hipStream_t st);   // the panels again from F.dev_W / F.dev_WT (same pattern, new values)
};

// out <- scale * A^-1 in on the unknowns' entries of the records (everything else in `out` is left alone); in != out
// class_of: the node set the roots' tile class is chosen for, if not mask.v (SpdSolverDev::fine_root_for)
void spd_run(int d, hipStream_t st, SpdSolverDev &S, NodeMask mask, double *in, double *out, double scale, const NodeBits *class_of = nullptr);

class Group {
 public:
  Group(const Graph &g, const std::vector<int> &node_ids, const Options &opt, int device);
  ~Group();
  bool ok() const { return ok_; }
  // the group cannot go on (a collective on its stream timed out, a refactorisation failed): update() / iterate() return -1
  // from now on and the destructor does not wait for the stream
  void mark_failed() { failed_ = true; }
  int d() const { return d_; }
  int num_local() const { return (int)nodes_.size(); }
  const DataInfo &info(int local) const { return info_[local]; }
  const NodeResults &results(int local) { finish_update(); return res_[local]; }
  const Options &options() const { return opt_; }
  int node_id(int local) const { return nodes_[local]; }

  // X: (d+1)(n0+n1) x d column-major with leading dimension ld (DPGOHash::initialize)
  int initialize(int local, const double *X, int ld);
  // X: global (d+1)N x d column-major; fills own and neighbour rows of every local node
  // (dist_pgo.cpp:435-446) and initialises them.
  int initialize_global(const double *X, int ld);
  // DPGOStar::evaluate_f / evaluate_grad at an arbitrary global X (DPGOStar.cpp:713-829); state untouched
  int evaluate_global(const double *X, int ld, double *F, double *grad_sqnorm, double *grad, int ldg);
  int set_options(const Options &o);                     // DPGOHash::set_options (DPGOHash.h:93-96)
  int update(const std::vector<int> &locals);
  int iterate(const std::vector<int> &locals);
  int communicate_local();
  // the driver's loop body (C++/examples/dist_pgo.cpp:496-521) for the nodes in `locals`: iterate -> exchange (the caller's:
  // an RCCL exchange on the communicator's stream, or none) -> communicate_local -> update.  Without an exchange the tail
  // of iterate() (Xk <- Xak) and the local halo copy are not launched on their own but become the head of update()'s first
  // segment (deferred_): one submission less per iteration where the host's launch rate is what bounds the group.
  int step(const std::vector<int> &locals, const std::function<int()> &exchange);
  // DPGOHash::receive (DPGOHash.cpp:45-82): msg for neighbour node beta is ((d+1) |recv[beta]|) x d,
  // [t rows ; R rows], poses in the order of recv[beta]; send() builds the message node `local` owes beta
  // from its current Xk (the poses of sent[beta], DPGO_utils.cpp:428-435)
  int receive(int local, int beta, const double *msg, int ld);
  int send(int local, int beta, double *msg, int ld) const;
  int num_recv(int local, int beta) const;
  int num_send(int local, int beta) const;

  // ---- distributed chordal initialisation (dchordal.cpp; C++/DChordal, C++/examples/dist_pgo.cpp:144-416).
  // Xlocal (optional): per-node local solutions in the global layout (stage 0); X: the initial guess, global
  // (d+1)N x d; objectives (optional): 0.5 sum |B X + b|^2 of the running stage every 20 iterations, stages in order.
  int dist_chordal_initialization(const DChordalOptions &o, const double *Xlocal, int ldl, double *X, int ld,
                                  std::vector<double> *objectives);
  // the sparse stages: setup -> initialize -> step ... -> get (DChordal_R / DChordal_t for all nodes in lockstep)
  int chordal_setup(int kind, double xi, const std::vector<std::vector<double>> &R);
  int chordal_initialize(const std::vector<std::vector<double>> &X);
  int chordal_step();
  double chordal_objective();
  int chordal_get(std::vector<std::vector<double>> &Xak);
  void chordal_release();

  // ---- AMM-PGO* (DPGOStar, C++/DPGO/src/DPGOStar.cpp:107-711); every node of the graph must be local
  int star_initialize_global(const double *X, int ld);   // DPGOStar::initialize (:107-124)
  int star_update();                                     // DPGOStar::update     (:306-313, update_n :315-390)
  int star_iterate();                                    // DPGOStar::iterate    (:126-213)
  double star_F() const { return starF_; }
  double star_fobj() const { return star_fobj_; }
  double star_fobjh() const { return star_fobjh_; }
  int star_branches() const { return star_branches_; }   // bit 0 pm, bit 1 mm, bit 2 phi fallback
  // boundary exchange across groups: records of the poses other groups need
  int num_sent() const { return (int)sent_rows_.size(); }
  // device buffer, num_sent()*RS doubles; st: the stream to enqueue on (default: the group's)
  int pack_sent(double *dev_buf, hipStream_t st = nullptr);
  // counts[r] keys of rank r (concatenated in nodes/poses); slot of key k of rank r = r*stride + k
  int set_recv_layout(int nranks, int stride, const int *counts, const int *nodes, const int *poses);
  int unpack_recv(const double *dev_gathered, hipStream_t st = nullptr);   // gathered buffer of all groups
  // a lazy unpack of buf through the lists dst (neighbour rows) / src (slots), device arrays (group.cpp); -1: not taken
  int set_pending_recv(const double *buf, int count, const int *dst_dev, const int *src_dev);
  void flush_pending_recv();
  // the exchange's pack rides on the tail of iterate(): rows (device) of the records to pack, how many, where to (null: off);
  // take_packed(): whether the last iterate() did it (and forgets it)
  void set_exchange_pack(const int *rows_dev, int n, double *dst) { pack_rows_ = rows_dev; pack_n_ = n; pack_dst_ = dst; packed_ = false; }
  bool take_packed() { const bool p = packed_; packed_ = false; return p; }
  // a wait for the group's stream ran into its deadline (a collective enqueued on it never ends): called before the error is raised
  void set_stuck_handler(void (*fn)(void *), void *user) { stuck_fn_ = fn; stuck_user_ = user; }
  // an exchange running on another stream (comm.cpp): `done` is recorded behind its unpack.  update() queues the
  // part of the surrogate build that needs no neighbour row, then makes the group's stream wait for it.
  void set_pending_exchange(hipEvent_t done) { xchg_done_ = done; }
  int device() const { return device_; }
  // AMM-PGO* across groups: the master's global objective needs the trial point's boundary poses of the other
  // groups (all-gather of `send` into `gathered`, stream-ordered on stream()) and sums of scalars over the groups
  typedef int (*AllGatherFn)(void *user);
  typedef int (*AllReduceFn)(void *user, double *vals, int n);
  int set_collectives(double *send_dev, double *gathered_dev, AllGatherFn ag, AllReduceFn ar, void *user);
  // optional: an in-place sum of n DEVICE doubles over the groups, enqueued on this group's stream (RCCL: comm.cpp).  With
  // it AMM-PGO*'s master sums never visit the host between the kernels that produce them and the one read-back.
  typedef int (*AllReduceDevFn)(void *user, double *dev_vals, int n);
  void set_device_allreduce(AllReduceDevFn f) { coll_allreduce_dev_ = f; }
  // results
  int get_Xk(int local, double *X, int ld) const;       // (d+1)(n0+n1) x d column-major
  int get_X_own(int local, double *X, int ld) const;    // Xak: (d+1) n0 x d
  int scatter_global(double *X, int ld) const;          // writes own poses into the global layout
  void sync() const;
  hipStream_t stream() const { return st_; }

  // test hooks (tests/ compare single operators with the oracle)
  int debug_apply(int local, const char *op, const double *in, int ld_in, double *out, int ld_out);
  const NodeOperators &host_ops(int local) const { return ops_[local]; }
  const SpdFactor &factor_tt() const { return Ltt_.F; }
  const SpdFactor &factor_rr() const { return Lrr_.F; }
  double lambda_max(int local) const { return lambda_max_[local]; }
  // (sent list across groups) unified own row of every boundary pose this group exports, with key
  const std::vector<std::pair<int, int>> &sent_keys() const { return sent_keys_; }
  const std::vector<int> &sent_rows() const { return sent_rows_; }
  // (node, pose) of every neighbour row whose owner lives in another group, and that row (unified numbering)
  void needed_keys(std::vector<std::pair<int, int>> &keys, std::vector<int> &rows) const;
  int num_records() const { return P0_ + P1_; }
  double *Xk_records() { return Xk_.p; }
  // dst[didx[k]] = src[sidx[k]] over pose records (didx may be null: dst[k]) on stream st
  void copy_records(hipStream_t st, int count, const int *didx, const int *sidx, const double *src, double *dst) const;

 private:
  bool ok_ = false;
  // A Dynamic rescale on the device commits its scales before the verdict on the new factor of G_tt is in (the host never
  // waits for it: rescale_device).  A non-positive pivot -- which the reference reports from inside its CHOLMOD call --
  // therefore leaves the group with operators it cannot solve with: it is marked failed and every later update() /
  // iterate() returns -1 (a new group is the way on; nothing is silently computed with a broken factor).
  mutable bool failed_ = false;
  int d_ = 0, RS_ = 0, B_ = 0, device_ = 0;
  Options opt_;
  std::vector<int> nodes_;
  std::map<int, int> local_of_node_;
  std::vector<DataInfo> info_;
  std::vector<NodeOperators> ops_;
  std::vector<NodeResults> res_;
  std::vector<double> lambda_max_;
  std::vector<std::map<int, int>> g_index_;   // per local node: local pose -> global pose
  int num_poses_global_ = 0, num_nodes_total_ = 1;
  // unified rows
  int P0_ = 0, P1_ = 0;
  std::vector<int> own_off_, nbr_off_;
  hipStream_t st_ = nullptr;

  // device data
  DevBuf<Seg> segs_;
  DevBuf<int> own_seg_ptr_, nbr_seg_ptr_;
  SegTable T_;
  // masks and per-node coefficients live in rings (device + pinned host), so changing them never
  // needs a stream synchronisation
  NodeMask cur_mask_ = ALL_NODES;  // the nodes the launches work on (set_mask), passed to the kernels by value
  double *h_scal_ = nullptr;       // pinned, written by k_reduce; the flag (one cache line further) follows the scalars
  unsigned long long *h_flag_ = nullptr, fetch_seq_ = 0;
  // a refactorisation of G_tt whose verdict (positive definite or not) has not been read yet: it is read at the first
  // wait for a read-back that was enqueued behind it (sequence number >= tt_verdict_seq_), or at sync()
  mutable bool tt_verdict_pending_ = false;
  unsigned long long tt_verdict_seq_ = 0;
  void check_tt_verdict(bool wait) const;
  double *h_cg_ = nullptr, *h_tnt_ = nullptr;   // pinned summaries of k_cg_scal / k_tnt_begin (same allocation as h_scal_)
  bool zc_ready_ = false;       // iterate() wrote Xk's own rows into the buffer the next update() rotates into X[iter]
  bool tnt_speculate_ = true;   // run_tnt: take the trial point behind the first CG step without waiting for its outcome
  // A whole CG step (Hessian product, both solves, the scalar kernels: ~20 launches with fixed arguments once masks, step
  // lengths and the flag's sequence number live on the device) captured as a HIP graph and replayed with ONE submission per
  // step (tnt.cpp).  For groups whose steps are bound by the host's launch rate (small graphs, one node per GPU);
  // DPGO_CG_GRAPH=0 / 1 forces it off / on.  Keyed by every pointer a step carries (the iterate's buffers swap).
  DevBuf<unsigned long long> dev_seq_;   // the device's copy of the last sequence number a kernel raised the flag to
  bool cg_graph_wanted() const;
  // ---- The branch-free SEGMENTS of an iteration as graph replays (round 5).  Between two read-backs an iteration is a fixed
  // sequence of launches -- update() behind the exchange, the start of iterate() up to the translation solve, a refinement
  // from its model gradient to the trial point's sums -- whose arguments are pointers, node sets and constants: everything
  // that changes from one iteration to the next lives in device memory (the Nesterov gammas: coefs_dev_, written by the one
  // eager launch of update(); the flag's sequence number: dev_seq_; the CG's masks and step lengths, as before).  segment()
  // runs such a sequence eagerly, or -- when the host's launch rate is what bounds the group (iter_graph_wanted) -- captures
  // it once per key (the buffers it touches, which rotate; the node set; the variant) and replays it with ONE submission.
  // Rare branches (a rejected step, a restart, a fallback, a Dynamic rescale) stay eager.  Bitwise the same results.
  // done_seq: once the read-back flag has reached it, the graph's last replay is over (it may be destroyed)
  struct SegGraph { std::vector<unsigned long long> key; hipGraphExec_t exec = nullptr; int flags = 0; unsigned long long used = 0, done_seq = 0; };
  std::vector<SegGraph> seg_graphs_;
  unsigned long long seg_clock_ = 0, graph_gen_ = 0;   // graph_gen_: bumped by whatever invalidates captured arguments
  bool capturing_ = false, graphs_broken_ = false;
  int captured_flags_ = 0;
  long seg_replays_ = 0, seg_captures_ = 0, seg_eager_ = 0, seg_captures_live_ = 0;   // _live_: since the last graphs_invalidate()
  bool capture_cap_warned_ = false;
  // DPGO_HOST_TIMING=1: where the host's time goes (seconds in hipGraphLaunch, in eagerly launched segments, in waits), on
  // stderr when the group goes
  bool host_timing_ = getenv("DPGO_HOST_TIMING") != nullptr;
  double t_graph_launch_ = 0, t_eager_seg_ = 0, t_wait_ = 0;
  long n_wait_ = 0, holes_total_ = 0, wait_hist_[6] = {0, 0, 0, 0, 0, 0};   // waits of < 50 us, < 200 us, < 1 ms, < 5 ms, < 50 ms, longer
  // the sequence number the next flag-raising launch carries: a fresh one, or 0 under capture (the kernel then takes the
  // device's count + 1, and the host counts along when the graph is replayed)
  unsigned long long next_seq() {
    if (capturing_) { captured_flags_++; return 0ull; }
    return ++fetch_seq_;
  }
  // ---- Whether segments are replayed (iter_graph_wanted), decided by MEASUREMENT: a group starts with eager launches and
  // keeps an eye on how much of the time it spends inside iterate() / update() is waiting for the GPU.  A host that waits most of the time (more than
  // 40 % of it) keeps up with eager launches, which are the faster way then (a replay costs the GPU ~8 us of start-up); a
  // host that waits less is what bounds the group -- a slow or busy box, a small graph whose kernels are shorter than a launch -- and its
  // segments are replayed from then on.  Looked at every 32 iterations; DPGO_ITER_GRAPH=0 / 1 forces either.
  bool host_bound_ = false;
  double win_wait_s_ = 0, win_lib_s_ = 0;   // of the window: seconds waiting for read-backs / seconds inside iterate() and update()
  int win_iters_ = 0;
  long win_nwait_ = 0, win_nlate_ = 0;   // of the window: waits for a read-back, and those that found it there already
  void host_bound_tick();        // once per iteration (update())
  struct InLib {                 // (the caller's own time between the calls is not the library's host being slow)
    Group *g; std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit InLib(Group *gg) : g(gg) {}
    ~InLib() { g->win_lib_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }
  };
  // launches that wait for the next segment to carry them (step()): captured pointer values, launched in order
  std::vector<std::function<void()>> deferred_;
  unsigned long long deferred_key_ = 0;
  bool defer_armed_ = false;
  void defer_or_launch(unsigned long long key, std::function<void()> fn);
  void flush_deferred();
  bool iter_graph_wanted() const;
  // bits: the nodes the sequence works on.  Only sequences over ALL the group's nodes are replayed: a partial set is a group
  // whose nodes are taking different branches, where the sets change from one iteration to the next and every new set
  // would be a new capture (measured: city10000 / 8 nodes with every subset captured ran 4 x slower than eagerly)
  // wanted: -1 = by iter_graph_wanted(), 0 / 1 = the caller's own policy (the CG steps: cg_graph_wanted)
  void segment(int id, NodeBits bits, std::initializer_list<unsigned long long> extra, const std::function<void()> &body,
               int wanted = -1);
  NodeBits all_bits() const { const int L = num_local(); return L >= 64 ? ~0ull : ((1ull << L) - 1); }
  void graphs_invalidate();   // waits (bounded) for the stream, destroys every captured graph, bumps graph_gen_
  void graphs_destroy();      // (the stream is known to be idle)
  bool drain(double seconds) const;
  DevBuf<double> coefs_dev_;  // per local node: gamma of the iteration under way (k_set_coefs)
 public:
  // (counters for the tests and the API-trace summary: replays, captures, segments run eagerly)
  void graph_stats(long *replays, long *captures, long *eager) const { *replays = seg_replays_; *captures = seg_captures_; *eager = seg_eager_; }
 private:
  // the mask of the nodes in `bits` (and-ed on the device with *p, if any) with the map that lets own-segment launches
  // cover these nodes only (kernels.h: NodeMask::nlive) when they are few
  NodeMask live_mask(NodeBits bits, const NodeBits *p) const;
  std::vector<int> own_seg_ptr_host_;
  // the fusions of round 6 (extrapolation and Dfobj inside the inter-edge pass, iterate()'s tail on the product with G, the
  // first CG step's vector update with the retraction): DPGO_FUSED=0 gives round 5's launch sequence (A/B hook; same bits)
  bool fused_ = true;
  // The tail of iterate() -- Xk <- Xak, and the buffer the next update() rotates into X[iter] -- waits for that update()'s
  // product with G, which reads the same records anyway and stores them on the way (k_bsr's copy1 / copy2): armed by step()
  // when no exchange stands between the two (the exchange's pack reads Xk); anything else launches it on its own.
  struct PendingTail { bool on = false; NodeMask m = ALL_NODES; const double *xak = nullptr; double *xk = nullptr, *z = nullptr; };
  PendingTail pending_tail_;
  // ---- The next update() enqueued AHEAD of the host's decision (round 6).  Between the trial point's read-back and the first
  // launch of update() the GPU used to idle for the host's acceptance test and bookkeeping (~13-17 us of an iteration of
  // 0.35-1.25 ms).  In the regime where that decision always comes out the same way -- every node refined, its CG over after
  // one step, the step accepted, no redo, no restart, no fallback: the whole early regime -- run_tnt() enqueues, right behind
  // the last kernel of the trial point and before it waits for it: the trial point's reduction with a GATE in the same launch
  // (k_reduce_gate) that takes that very decision on the device from the same sums, and the launches the common course leads to -- the local halo copy, update()'s product with G
  // (which carries iterate()'s tail) and its inter-edge pass and reduction, with the buffers in the roles they will have
  // after the accepted step and the rotation of the history -- under the device word the gate sets.  The host then takes its
  // decision as ever; if it is the common one, iterate() / communicate_local() / update() find their launches done and only
  // do their bookkeeping; if not, those launches have fallen through (they touched nothing) and the normal path runs.  The
  // two verdicts come from the same bits through the same operations; the host checks that they agree when it next waits
  // (finish_update).  Armed by step() without an exchange; eager launches only; DPGO_SPEC_UPDATE=0 switches it off.
  struct SpecUpdate {
    bool on = false, consumed_copy = false;
    unsigned long long seq_trial = 0, seq_last = 0;  // the flags of the trial point's reduction (+ gate) and of the continuation's last flag-raising launch
    bool lazy = false;                               // the continuation left update()'s reduction to the next refinement (UpdLazy)
    const double *xak = nullptr; double *zc = nullptr, *gc = nullptr, *dfc = nullptr, *gx = nullptr;   // the roles it was enqueued with
  };
  SpecUpdate spec_upd_;
  // update()'s closing reduction is only read by the host, late (finish_update): where the read-back is deferred the
  // reduction is not launched at all but rides on the next refinement's k_cg_scal_begin (one workgroup per node anyway),
  // from partial-sum slots of its own; if no refinement comes, finish_update() launches it
  struct UpdLazy { bool pending = false; int nslots = 0; };
  UpdLazy upd_lazy_;
  bool lazy_update_reduce() const;
  long n_spec_enqueued_ = 0, n_spec_stood_ = 0;   // (DPGO_HOST_TIMING=1 prints them)
  bool spec_update_armed_ = false, spec_update_enabled_ = true;
  bool tnt_common_ = false;          // run_tnt: the refinement took the common course (one step, accepted by every node, over)
  DevBuf<NodeBits> go_;              // the gate's word
  DevBuf<double> dev_sums_, dev_tnt_;   // the trial point's sums / the refinement's start, per node, in device memory
  double *h_gate_ = nullptr;         // pinned (same allocation as h_scal_): the gate's verdict
  bool spec_verdict_pending_ = false, spec_verdict_expected_ = false;   // the host acted on its own verdict; the gate's is compared at the next wait
  unsigned long long spec_verdict_seq_ = 0;
  bool spec_update_possible(const double *xprop) const;
  void speculate_update(const double *xprop, int nslots_trial);   // run_tnt: the trial point's reduction + gate, then the continuation
  void check_gate(bool host_common);
  bool tail_fusable_ = false;
  void flush_pending_tail();
  DevBuf<unsigned> reduce_arrived_;
  DevBuf<double> partials_;
  DevBuf<CgNode> cg_;       // device-resident state of the truncated CG (tnt.cpp, k_cg_scal)
  DevBuf<double> jacobi_;   // Preconditioner::Jacobi: 1 / diag(G_RR), one entry per rotation row
  DevBuf<NodeBits> dmask_;  // [0] nodes taking the next Hessian product, [1] nodes going on to the preconditioner, [2] nodes whose CG is over
  struct BsrBufs { DevBuf<int> ptr, col; DevBuf<double> val, tcol; BsrDev dev; };   // tcol: first column of every block (G only)
  BsrBufs G_, S_, P_, P0m_, Q_;
  DevBuf<double> Dd_, Qd_, Tinv_, N_, V_;
  DevBuf<int> e_tail_, e_head_, e_inc_ptr_, e_inc_;
  DevBuf<double> e_R_, e_t_, e_kappa_, e_tau_;
  DevBuf<InterInc> e_rec_;        // the same data once more, one record per incidence (k_inter)
  InterEdgesDev E_;
  DevBuf<int> i_tail_, i_head_, i_inc_ptr_, i_inc_;
  DevBuf<double> i_R_, i_t_, i_kappa_, i_tau_;
  InterEdgesDev Ei_;              // intra-node edges in residual form (objective evaluation only)
  SpdSolverDev Ltt_, Lrr_;
  // halo
  DevBuf<int> gather_dst_, gather_src_;   // local halo copy lists
  std::vector<int> sent_rows_;     // unified own rows exported to other groups
  std::vector<std::pair<int, int>> sent_keys_;   // (node, pose) of each exported row
  DevBuf<int> sent_rows_dev_;
  DevBuf<int> recv_dst_, recv_src_;       // remote halo: nbr row <- gathered slot
  const double *pending_recv_ = nullptr;  // a lazy unpack nobody has consumed yet (set_pending_recv)
  const int *recv_key_ = nullptr, *recv_dst_dev_ = nullptr, *recv_src_dev_ = nullptr;
  int recv_count_ = 0;
  DevBuf<int> recv_nsrc_;                 // per neighbour row: its slot in the receive buffer, -1: none
  std::vector<InterInc> e_rec_host_;      // host copy of the incidence records (their osrc field follows the receive lay-out)
  const int *pack_rows_ = nullptr;
  int pack_n_ = 0;
  double *pack_dst_ = nullptr;
  bool packed_ = false;
  void (*stuck_fn_)(void *) = nullptr;
  void *stuck_user_ = nullptr;
  // vectors (records)
  DevBuf<double> Xk_, Zc_, Zp_, Y_, DfE_, Tall_;                 // P0+P1 rows
  DevBuf<double> Xak_, Xakh_, gc_, gp_, Dfc_, Dfp_, gx_, Dfx_, T1_;   // P0 rows
  // robust loss, Static rescale: G X[k] and G X[k-1] (own rows), rotated with the history of X: the product with G at the
  // extrapolated point is their linear combination (prepare_extrapolated), one pass over the operator less per iteration
  DevBuf<double> GXc_, GXp_;
  bool keep_gx() const { return opt_.loss != 0 && !dynamic(); }
  DevBuf<double> tmp_[14];                                       // P0 rows, TNT work vectors
  hipEvent_t xchg_done_ = nullptr;   // pending boundary exchange (not owned)
  void join_exchange();              // the group's stream waits for it
  struct ChordalState;
  ChordalState *ch_ = nullptr;
  bool star_ = false;
  double *coll_send_ = nullptr, *coll_gathered_ = nullptr;
  AllGatherFn coll_allgather_ = nullptr;
  AllReduceFn coll_allreduce_ = nullptr;
  AllReduceDevFn coll_allreduce_dev_ = nullptr;
  DevBuf<double> star_vals_;   // AMM-PGO*: the master's four sums on the device (k_star_sums)
  void *coll_user_ = nullptr;
  double starF_ = 0, star_fobj_ = 0, star_fobjh_ = 0;
  int star_branches_ = 0;
  void node_rows_of_global(int a, const double *X, int ld, std::vector<double> &Z) const;
  bool prepare_extrapolated(const double *gam_dev = nullptr, int prox_slot = -1);   // Y, g_x, Df_x for the masked nodes (+ the proximal step)
  double global_objective(const double *X_own);           // F at the point whose own rows are X_own
  // the master's numbers in ONE read-back: F(X1) [, F(X2)] [, |X1 - ref|^2, |X2 - ref|^2] (null pointers: not wanted)
  int star_sums(const double *X1_own, const double *X2_own, const double *ref_own, double *F1, double *F2, double *d1, double *d2);
  void enqueue_objective(const double *X_own, int slot0);   // k_cost partials of the point into slots slot0, slot0 + 1

  void upload_bsr(const std::vector<const BsrMatrix *> &per_node, bool rows_all, BsrBufs &out);
  void upload_operators();
  int refactor_tt();
  // Rescale::Dynamic (DPGOProblem.cpp:289-358, 426-514, 751-840): scale of every inter-node edge per node, the
  // counter of DPGOResult::rescale_count, the edge weights of the last evaluate_E
  std::vector<std::vector<double>> scale_;
  std::vector<int> rescale_count_;
  std::vector<int> e_off_;          // first inter edge of every node in E_
  DevBuf<double> e_w_;
  bool dynamic() const { return opt_.loss != 0 && opt_.rescale == 1; }
  // ... on the device (DPGO_RESCALE_HOST=1 keeps the host path: weights read back, operators re-assembled and uploaded):
  // the scales, counters and decisions live in device memory, the block-diagonal terms are rebuilt by k_rescale_apply
  // and G_tt is re-factored from values that never leave the GPU (SpdFactor::keep_numeric)
  bool device_rescale_ = false;
  DevBuf<double> e_scale_, Gbase_, Hbase_;
  DevBuf<int> rs_count_, rs_flags_, gpos_, att_pos_, e_off_dev_;
  double *h_rs_ = nullptr;   // pinned (same allocation as h_scal_): the rescale decision of every node
  std::vector<int> rescale_device(const std::vector<int> &set);   // after the decision arrived: apply + refactor; returns the rescaled nodes
  void setup_device_rescale();
  // decide per node whether its surrogate is rescaled (weights in e_w_), rebuild what changed; returns the nodes
  // that were rescaled
  std::vector<int> maybe_rescale(const std::vector<int> &set);
  void set_mask(const std::vector<int> &locals);
  void fetch(int nslots, bool all_rows);                  // -> h_scal_[local * MAX_SLOTS + s]
  void wait_flag(unsigned long long seq);
  int deferred_slots_ = 0;   // slots written earlier that ride along with the next fetch (saves a host round trip)
  double scal(int local, int s) const { return h_scal_[local * MAX_SLOTS + s]; }
  double *h_upd_ = nullptr;   // pinned (same allocation): the sums update() ends with
  double uscal(int local, int s) const { return h_upd_[local * MAX_SLOTS + s]; }
  bool spec_refined_ = false; // amm(): every node of the group was refined in the last iteration (the next one starts its refinement unasked)
  void copy_rows(double *dst, const double *src, bool all_rows, int part = 0);
  // (a captured CG step launches over every node but takes the tile classes of the eager step's node sets: tnt.cpp, graph_step)
  const NodeBits *class_tt_ = nullptr, *class_rr_ = nullptr;
  void solve_tt(double *in, double *out, double scale);   // out.t <- scale * G_tt^-1 in.t
  void solve_rr(double *in, double *out, double scale);   // out.R <- scale * (G_RR + lambda I)^-1 in.R
  void apply_tcol(const double *xt, const double *base, double *y, int mode = 0, const double *X = nullptr,
                  const double *nabla = nullptr, const double *Rdot = nullptr, double *out2 = nullptr,
                  const double *rres = nullptr, double *partials = nullptr, const double *dg = nullptr,
                  const double *dga = nullptr, const double *ds = nullptr, const double *dgrad = nullptr,
                  const double *dhs = nullptr);
  void recover_translations(double *X, const double *g);  // X.t = -Gtt^-1 (g_t + G_tR X.R) for masked nodes
  void eval_G(const double *X, const double *g, int slot);
  void host_update_logic(int local, double fobj, double f, double gradFnorm);
  // The read-back that ends update() is deferred where nothing has to be decided yet: update() enqueues the reduction,
  // advances the Nesterov sequence (host_update_pre: s, gamma -- they do not depend on the numbers read back) and
  // returns; the next iterate() queues its extrapolation, proximal step and translation solve and only then waits
  // (finish_update), so the GPU does not idle while the host takes the scalars.  Every other reader of the node state
  // or of the pinned scalars calls finish_update() first.
  void host_update_pre(int local);
  void finish_update();
  unsigned long long fetch_async(int nslots, bool all_rows);
  std::function<void()> pending_update_;
  unsigned long long pending_seq_ = 0;
  int amm(const std::vector<int> &locals);
  int mm(const std::vector<int> &locals);
  // refine X in place (TNT on G(. | g)); sets Gk = G(X | g) and, with g_alt, Gk_alt = G(X | g_alt)
  // base_ready: X.t was just recovered from X.R with this g (recover_translations) and T1_ still holds that product
  // confirm (optional): called once the start of the refinement is enqueued -- the caller's chance to take a read-back
  // that decides whether these nodes are refined at all; false: the refinement is abandoned (returns false; what was
  // enqueued only touched work vectors, and T1_)
  bool run_tnt(const std::vector<int> &locals, double *X, const double *g, const double *g_alt = nullptr,
               bool base_ready = false, const std::function<bool()> *confirm = nullptr);
};

}  // namespace dpgo
