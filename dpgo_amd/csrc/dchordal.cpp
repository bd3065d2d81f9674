// Distributed chordal initialisation: the `--dist_init true` branch of the reference driver
// (C++/examples/dist_pgo.cpp:144-416) built on C++/DChordal.
//
// Four Nesterov-accelerated MM stages on linear least-squares problems:
//   1 DChordalReduced_R  one d x d block per node        (DChordalReduced.cpp:114-183, DChordal_utils.cpp:67-309)
//   2 DChordal_R         rotations of all poses, relaxed (DChordal.cpp:79-152, DChordal_utils.cpp:605-913)
//   3 DChordalReduced_t  one translation per node        (DChordal_utils.cpp:311-603)
//   4 DChordal_t         translations of all poses       (DChordal_utils.cpp:915-1204)
// Every iteration is Y = (1 + gamma) X_k - gamma X_{k-1}, X^a <- -G^-1 (g_ + S Y), halo copy.  Stages 2 and 4 are the
// sparse ones: they run on the device with the kernels of the main path -- k_extrapolate, k_bsr (S Y + g_), the
// multifrontal SPD solve (k_spd_level) on the factor of G, k_copy_indexed for the halo -- on the pose records of
// the group (rotation rows for stage 2, translation row for stage 4).  Stages 1 and 3 have d x d / scalar unknowns
// per node and a handful of flops per iteration: host.
//
// Stage 0 of the reference is a per-node SE-Sync solve (DChordal_utils.cpp:11-28), a third-party solver that is
// out of scope; its stand-in is the chordal initialisation of the node's intra-node subgraph followed by
// `local_iters` iterations of MM-PGO with the truncated-Newton refinement forced on, run on the device for all
// nodes at once (a group over the graph without its inter-node edges).  A caller that has local solutions from
// elsewhere passes them in (X_local).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <set>

#include "group.h"

namespace dpgo {

int chordal_initialization(const Graph &g, double *X, int ld);
void project_to_SOd_host(int d, double *M);

namespace {
#define HIP_OK(x)                                                                                  \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      fprintf(stderr, "[dpgo_amd] ERROR: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      throw DeviceError(hipGetErrorString(e_));                                                    \
    }                                                                                              \
  } while (0)

// C (d x d) = A^T B, C = A B, C = A B^T on row-major d x d blocks
void mul_tn(int d, const double *A, const double *B, double *C) {
  for (int r = 0; r < d; r++)
    for (int c = 0; c < d; c++) {
      double a = 0;
      for (int k = 0; k < d; k++) a += A[k * d + r] * B[k * d + c];
      C[r * d + c] = a;
    }
}
void mul_nn(int d, const double *A, const double *B, double *C) {
  for (int r = 0; r < d; r++)
    for (int c = 0; c < d; c++) {
      double a = 0;
      for (int k = 0; k < d; k++) a += A[r * d + k] * B[k * d + c];
      C[r * d + c] = a;
    }
}
void mul_nt(int d, const double *A, const double *B, double *C) {
  for (int r = 0; r < d; r++)
    for (int c = 0; c < d; c++) {
      double a = 0;
      for (int k = 0; k < d; k++) a += A[r * d + k] * B[c * d + k];
      C[r * d + c] = a;
    }
}
// inverse of a small SPD matrix (d <= 3) by Gauss-Jordan
void inv_small(int d, const double *A, double *Ai) {
  double M[3][6];
  for (int r = 0; r < d; r++)
    for (int c = 0; c < d; c++) { M[r][c] = A[r * d + c]; M[r][d + c] = (r == c); }
  for (int p = 0; p < d; p++) {
    const double piv = M[p][p];
    for (int c = 0; c < 2 * d; c++) M[p][c] /= piv;
    for (int r = 0; r < d; r++)
      if (r != p) {
        const double f = M[r][p];
        for (int c = 0; c < 2 * d; c++) M[r][c] -= f * M[p][c];
      }
  }
  for (int r = 0; r < d; r++)
    for (int c = 0; c < d; c++) Ai[r * d + c] = M[r][d + c];
}

// The Nesterov sequence every stage shares (DChordal.cpp:110-121): s_0 = 1, gamma_k = (s_k - 1) / s_{k+1}
struct Nesterov {
  double s = 1.0;
  double next_gamma() {
    const double s1 = 0.5 + 0.5 * std::sqrt(4.0 * s * s + 1.0);
    const double g = (s - 1.0) / s1;
    s = s1;
    return g;
  }
};

// One node's reduced problem: unknown block of p rows (p = d: rotation, p = 1: translation), neighbours' blocks
// behind it in n_index order (DChordal_utils.cpp:47-60)
struct ReducedNode {
  int a = 0, nn = 0, p = 1, d = 3;
  std::map<int, int> n_index;
  std::vector<double> Ginv;          // p x p (rotation) or 1 x 1
  std::vector<double> S;             // p x (nn+1) p   (rotation: d x d blocks; translation: scalars)
  std::vector<double> g;             // p x d constant
  std::vector<double> Bm, b;         // residual rows for the objective: rotation (M d) x ((nn+1) d), translation M x (nn+1)
  int M = 0;
  std::vector<double> Xk, Xc, Xp;    // (nn+1) p x d : current, X[k], X[k-1]
  Nesterov nes;
  void initialize(const std::vector<double> &X) { Xk = Xc = Xp = X; nes = Nesterov(); }
  void update() { Xp = Xc; Xc = Xk; }
  void iterate() {
    const double gam = nes.next_gamma();
    const int rows = (nn + 1) * p;
    std::vector<double> Y((size_t)rows * d), gg((size_t)p * d);
    for (size_t k = 0; k < Y.size(); k++) Y[k] = Xc[k] + gam * (Xc[k] - Xp[k]);
    for (int r = 0; r < p; r++)
      for (int c = 0; c < d; c++) {
        double acc = g[r * d + c];
        for (int k = 0; k < rows; k++) acc += S[(size_t)r * rows + k] * Y[(size_t)k * d + c];
        gg[r * d + c] = acc;
      }
    for (int r = 0; r < p; r++)
      for (int c = 0; c < d; c++) {
        double acc = 0;
        for (int k = 0; k < p; k++) acc += Ginv[r * p + k] * gg[k * d + c];
        Xk[r * d + c] = -acc;
      }
  }
  double objective() const {   // |B Xk + b|^2
    const int rows = (nn + 1) * p, mr = (int)(b.size() / d);
    double f = 0;
    for (int r = 0; r < mr; r++)
      for (int c = 0; c < d; c++) {
        double acc = b[(size_t)r * d + c];
        for (int k = 0; k < rows; k++) acc += Bm[(size_t)r * rows + k] * Xk[(size_t)k * d + c];
        f += acc * acc;
      }
    return f;
  }
};

std::map<int, int> make_n_index(const DataInfo &info) {
  std::set<int> nb;
  for (const auto &k : info.nbr_key) nb.insert(k.first);
  std::map<int, int> out;
  out[info.node] = 0;
  int cnt = 1;
  for (int b : nb) out[b] = cnt++;
  return out;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// The sparse stages on the device (members of Group: they use its rows, segments, halo lists and buffers)
// ---------------------------------------------------------------------------------------------------------
struct Group::ChordalState {
  int kind = -1;           // 0 rotations (DChordal_R), 1 translations (DChordal_t)
  SpdSolverDev L;
  BsrBufs S;
  DevBuf<double> gconst;
  Nesterov nes;
  std::vector<std::vector<double>> Rfix;   // kind 1: the fixed rotations Y of own + neighbour poses, per node
};

void Group::chordal_release() {
  delete ch_;
  ch_ = nullptr;
}

// kind 0: G = intra connection Laplacian + 2 kappa per inter edge on the local endpoint (xi cancels,
// DChordal_utils.cpp:891-894); node 0 pins its first block (DChordalProblem.cpp:60-62).  kind 1: G = tau-Laplacian
// + 2 tau per inter edge + xi, g_ from the fixed rotations R (per node (n0 + n1) d x d blocks, Y = R^T as everywhere).
int Group::chordal_setup(int kind, double xi, const std::vector<std::vector<double>> &R) {
  chordal_release();
  zc_ready_ = false;   // (the stages use the history buffers as their own)
  ch_ = new ChordalState();
  ch_->kind = kind;
  const int L = num_local(), d = d_, B = B_;
  CsrMatrix A;
  A.ptr.push_back(0);
  std::vector<BsrMatrix> Sn(L);
  std::vector<double> gc((size_t)P0_ * RS_, 0.0);
  const int dof = kind == 0 ? d : 1;
  std::vector<int> node_of_unknown;
  for (int a = 0; a < L; a++) {
    const DataInfo &info = info_[a];
    const int n0 = info.n[0], n1 = info.n[1];
    const bool pin = kind == 0 && nodes_[a] == 0;
    // dense-ish assembly through maps: (row unknown) -> {col unknown -> value}
    std::vector<std::map<int, double>> rows((size_t)n0 * dof);
    std::vector<std::map<int, std::vector<double>>> srows(n0);   // S: block row -> {unified local col -> (d+1)^2 block}
    auto sblk = [&](int r, int c) -> std::vector<double> & {
      auto &v = srows[r][c];
      if (v.empty()) v.assign((size_t)B * B, 0.0);
      return v;
    };
    double *g = gc.data() + (size_t)own_off_[a] * RS_;
    if (kind == 0) {
      for (const auto &m : info.intra) {
        const int i = info.tail(m), j = info.head(m);
        for (int k = 0; k < d; k++) { rows[i * d + k][i * d + k] += m.kappa; rows[j * d + k][j * d + k] += m.kappa; }
        for (int r = 0; r < d; r++)
          for (int c = 0; c < d; c++) {
            rows[i * d + r][j * d + c] += -m.kappa * m.R[r * d + c];
            rows[j * d + r][i * d + c] += -m.kappa * m.R[c * d + r];
          }
      }
      for (const auto &m : info.inter) {
        const int i = info.tail(m), j = info.head(m);   // unified within the node: own < n0 <= neighbour
        const bool tail_local = m.inode == nodes_[a];
        const int own = tail_local ? i : j, nbr = tail_local ? j : i;
        for (int k = 0; k < d; k++) rows[own * d + k][own * d + k] += 2 * m.kappa;
        auto &sd = sblk(own, own);
        auto &so = sblk(own, nbr);
        for (int r = 0; r < d; r++) {
          sd[(1 + r) * B + 1 + r] -= m.kappa;
          for (int c = 0; c < d; c++) so[(1 + r) * B + 1 + c] -= m.kappa * (tail_local ? m.R[r * d + c] : m.R[c * d + r]);
        }
      }
      if (pin) {
        // the first block stays what it is: identity rows in G, S row = -I (so that -G^-1 (S Y) = Y there), and the
        // couplings of the other rows to it move to g_ = G(rest, first) I   (DChordalProblem.cpp:60-62)
        for (int r = 0; r < d; r++) {
          for (int i = d; i < n0 * d; i++) {
            auto it = rows[i].find(r);
            if (it != rows[i].end()) {
              g[(size_t)(i / d) * RS_ + d + (i % d) * d + r] += it->second;   // times Y_first = I: column r
              rows[i].erase(it);
            }
          }
          rows[r].clear();
          rows[r][r] = 1.0;
        }
        srows[0].clear();
        auto &sd = sblk(0, 0);
        for (int r = 0; r < d; r++) sd[(1 + r) * B + 1 + r] = -1.0;
      }
    } else {
      if ((int)R.size() != L || (int)R[a].size() != (n0 + n1) * d * d) return -1;
      ch_->Rfix = R;
      auto nt_of = [&](const Measurement &m, int tail, double *nt) {   // nt = Y_tail^T t_e
        const double *Y = &R[a][(size_t)tail * d * d];
        for (int c = 0; c < d; c++) {
          double acc = 0;
          for (int k = 0; k < d; k++) acc += Y[k * d + c] * m.t[k];
          nt[c] = acc;
        }
      };
      double nt[3];
      for (const auto &m : info.intra) {
        const int i = info.tail(m), j = info.head(m);
        rows[i][i] += m.tau; rows[j][j] += m.tau; rows[i][j] -= m.tau; rows[j][i] -= m.tau;
        nt_of(m, i, nt);
        for (int c = 0; c < d; c++) { g[(size_t)i * RS_ + c] += m.tau * nt[c]; g[(size_t)j * RS_ + c] -= m.tau * nt[c]; }
      }
      for (const auto &m : info.inter) {
        const int i = info.tail(m), j = info.head(m);
        const bool tail_local = m.inode == nodes_[a];
        const int own = tail_local ? i : j, nbr = tail_local ? j : i;
        rows[own][own] += 2 * m.tau;
        sblk(own, own)[0] -= m.tau;
        sblk(own, nbr)[0] -= m.tau;
        nt_of(m, i, nt);
        for (int c = 0; c < d; c++) g[(size_t)own * RS_ + c] += (tail_local ? 1.0 : -1.0) * m.tau * nt[c];
      }
      for (int i = 0; i < n0; i++) {
        rows[i][i] += xi;
        sblk(i, i)[0] -= xi;
      }
    }
    const int base = own_off_[a] * dof;
    for (size_t r = 0; r < rows.size(); r++) {
      for (const auto &cv : rows[r]) { A.col.push_back(base + cv.first); A.val.push_back(cv.second); }
      A.ptr.push_back((int)A.col.size());
      node_of_unknown.push_back(a);
    }
    BsrMatrix &S = Sn[a];
    S.B = B; S.nrows = n0; S.ncols = n0 + n1;
    S.ptr.assign(1, 0);
    for (int r = 0; r < n0; r++) {
      for (const auto &cv : srows[r]) {
        S.col.push_back(cv.first);
        S.val.insert(S.val.end(), cv.second.begin(), cv.second.end());
      }
      S.ptr.push_back((int)S.col.size());
    }
  }
  A.n = (int)A.ptr.size() - 1;
  if (spd_factor(A, ch_->L.F, 64, 0, dof, true) != 0) {   // (ordering on the pose graph, factor kept on the device)
    fprintf(stderr, "[dpgo_amd] ERROR: distributed chordal initialisation: the stage matrix is not positive definite "
                    "(a node without inter-node edges, or a node whose poses are not connected?).\n");
    return -1;
  }
  ch_->L.dof = dof;
  ch_->L.upload(d, node_of_unknown);
  std::vector<const BsrMatrix *> v(L);
  for (int a = 0; a < L; a++) v[a] = &Sn[a];
  upload_bsr(v, false, ch_->S);
  ch_->gconst.upload(gc);
  return 0;
}

// X[a]: (n0 + n1) blocks (kind 0: d x d, kind 1: 1 x d), own poses first (DChordal::initialize, DChordal.cpp:8-43)
int Group::chordal_initialize(const std::vector<std::vector<double>> &X) {
  if (!ch_) return -1;
  const int d = d_, bs = ch_->kind == 0 ? d * d : d, off = ch_->kind == 0 ? d : 0;
  std::vector<double> rec((size_t)(P0_ + P1_) * RS_, 0.0);
  for (int a = 0; a < num_local(); a++) {
    const int n0 = info_[a].n[0], n1 = info_[a].n[1];
    if ((int)X[a].size() != (n0 + n1) * bs) return -1;
    for (int k = 0; k < n0 + n1; k++) {
      const size_t row = k < n0 ? own_off_[a] + k : P0_ + nbr_off_[a] + (k - n0);
      std::copy(&X[a][(size_t)k * bs], &X[a][(size_t)(k + 1) * bs], &rec[row * RS_ + off]);
    }
  }
  sync();
  for (double *dst : {Xk_.p, Zc_.p, Zp_.p}) HIP_OK(hipMemcpy(dst, rec.data(), sizeof(double) * rec.size(), hipMemcpyHostToDevice));
  ch_->nes = Nesterov();
  return 0;
}

// update(); iterate(); communicate() of every node (DChordal.cpp:79-152, DChordal.h:26-84)
int Group::chordal_step() {
  if (!ch_) return -1;
  std::vector<int> all(num_local());
  for (int a = 0; a < num_local(); a++) all[a] = a;
  set_mask(all);
  Zp_.swap(Zc_);                       // X[k-1] <- X[k]
  copy_rows(Zc_.p, Xk_.p, true);       // X[k] = Xk
  NodeCoefs gam;
  const double gm = ch_->nes.next_gamma();
  for (int a = 0; a < MAX_LOCAL_NODES; a++) gam.a[a] = gam.b[a] = gm;
  launch_extrapolate(d_, st_, T_, true, cur_mask_, gam, Zc_.p, Zp_.p, Y_.p);
  launch_bsr(d_, st_, T_, false, cur_mask_, ch_->S.dev, Y_.p, false, ch_->gconst.p, T1_.p, nullptr, 0, nullptr, nullptr, 0);
  spd_run(d_, st_, ch_->L, cur_mask_, T1_.p, Xk_.p, -1.0);   // Xak = -G^-1 (g_ + S Y), straight into Xk's own rows
  if (communicate_local() != 0) return -1;
  if (coll_allgather_) {   // neighbours hosted by other groups: the boundary rows travel like the iterate's (DChordal.h:26-84)
    launch_copy_indexed(d_, st_, (int)sent_rows_.size(), nullptr, sent_rows_dev_.p, Xk_.p, coll_send_);
    if (coll_allgather_(coll_user_) != 0) { fprintf(stderr, "[dpgo_amd] ERROR: all-gather callback failed.\n"); return -1; }
    launch_copy_indexed(d_, st_, (int)recv_dst_.n, recv_dst_.p, recv_src_.p, coll_gathered_, Xk_.p);
  }
  return 0;
}

// 0.5 * sum_a |B Xk + b|^2 (DChordal_utils.h:129-140), on the host from a copy of Xk
double Group::chordal_objective() {
  if (!ch_) return NAN;
  sync();
  std::vector<double> rec((size_t)(P0_ + P1_) * RS_);
  HIP_OK(hipMemcpy(rec.data(), Xk_.p, sizeof(double) * rec.size(), hipMemcpyDeviceToHost));
  const int d = d_;
  double f = 0;
  for (int a = 0; a < num_local(); a++) {
    const DataInfo &info = info_[a];
    const int n0 = info.n[0];
    auto row = [&](int k) { return &rec[(size_t)(k < n0 ? own_off_[a] + k : P0_ + nbr_off_[a] + (k - n0)) * RS_]; };
    for (int pass = 0; pass < 2; pass++)
      for (const auto &m : pass == 0 ? info.intra : info.inter) {
        const int i = info.tail(m), j = info.head(m);
        const double *zi = row(i), *zj = row(j);
        if (ch_->kind == 0) {
          for (int r = 0; r < d; r++)
            for (int c = 0; c < d; c++) {
              double acc = -zj[d + r * d + c];
              for (int k = 0; k < d; k++) acc += m.R[k * d + r] * zi[d + k * d + c];
              f += m.kappa * acc * acc;
            }
        } else {
          const double *Y = &ch_->Rfix[a][(size_t)i * d * d];
          for (int c = 0; c < d; c++) {
            double acc = zi[c] - zj[c];
            for (int k = 0; k < d; k++) acc += Y[k * d + c] * m.t[k];
            f += m.tau * acc * acc;
          }
        }
      }
  }
  return 0.5 * f;
}

// results().Xak of every node: n0 blocks
int Group::chordal_get(std::vector<std::vector<double>> &Xak) {
  if (!ch_) return -1;
  sync();
  const int d = d_, bs = ch_->kind == 0 ? d * d : d, off = ch_->kind == 0 ? d : 0;
  std::vector<double> rec((size_t)P0_ * RS_);
  HIP_OK(hipMemcpy(rec.data(), Xk_.p, sizeof(double) * rec.size(), hipMemcpyDeviceToHost));
  Xak.assign(num_local(), {});
  for (int a = 0; a < num_local(); a++) {
    const int n0 = info_[a].n[0];
    Xak[a].resize((size_t)n0 * bs);
    for (int k = 0; k < n0; k++)
      std::copy(&rec[(size_t)(own_off_[a] + k) * RS_ + off], &rec[(size_t)(own_off_[a] + k) * RS_ + off + bs], &Xak[a][(size_t)k * bs]);
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// The driver schedule (dist_pgo.cpp:144-416)
// ---------------------------------------------------------------------------------------------------------
// The nodes of the graph may be spread over several groups (one per GPU / process) connected by the collectives of
// dpgo_group_set_collectives / dpgo_comm_create: every group runs the schedule for ITS nodes, and what a node needs from
// another one travels the way the reference's halo exchanges do (DChordal.h:26-84, DChordalReduced.h:24-51) --
//   * per iteration of a sparse stage: the boundary rows of Xk, the device all-gather of the main exchange (chordal_step);
//   * per iteration of a reduced stage: every node's own block, a sum over the groups of an array with one slot per node;
//   * between the stages: own-pose blocks by global pose id / per-node blocks by node id, summed the same way (each
//     entry has exactly one owner, so the sum is a gather and every group ends up with the same bits);
//   * the sampled objectives: sums over the groups.
// With one group that hosts every node the same code runs on local copies.
int Group::dist_chordal_initialization(const DChordalOptions &o, const double *Xlocal, int ldl, double *X, int ld,
                                       std::vector<double> *objectives) {
  finish_update();
  const int N = num_local(), Nn = num_nodes_total_, d = d_, NP = num_poses_global_;
  SetupClock clk;   // (DPGO_SETUP_TIMING=1)
  const bool spread = N != Nn;
  if (spread && !(coll_allreduce_ && coll_allgather_)) {
    fprintf(stderr, "[dpgo_amd] ERROR: the distributed chordal initialisation needs every node of the graph in the group, or "
                    "collectives (dpgo_group_set_collectives / dpgo_comm_create) that connect the groups.\n");
    return -1;
  }
  if (Nn < 2) {   // the reference reads inter_measurements[0] unguarded (DChordal_utils.cpp:86,383,627,937)
    fprintf(stderr, "[dpgo_amd] ERROR: the distributed chordal initialisation needs at least two nodes (use the centralised one).\n");
    return -1;
  }
  if (ld < (d + 1) * NP || (Xlocal && ldl < (d + 1) * NP)) return -1;
  for (int a = 0; a + 1 < N; a++)
    if (nodes_[a] >= nodes_[a + 1]) {
      fprintf(stderr, "[dpgo_amd] ERROR: the distributed chordal initialisation expects the nodes in order.\n");
      return -1;
    }
  if (objectives) objectives->clear();
  auto gid = [&](int a, int k) { return g_index_[a].at(info_[a].own_pose[k]); };
  // global id of a neighbour pose from the partition rule (DPGO_utils.cpp:147-158)
  const int pq = NP / Nn, pinc = NP - Nn * pq;
  auto key_gid = [&](const std::pair<int, int> &key) {
    return (key.first < pinc ? key.first * (pq + 1) : pinc * (pq + 1) + (key.first - pinc) * pq) + key.second;
  };
  auto allsum = [&](double *v, size_t n) -> int {
    if (!coll_allreduce_) return 0;
    for (size_t off = 0; off < n; off += (size_t)1 << 28) {
      const int m = (int)std::min<size_t>((size_t)1 << 28, n - off);
      if (coll_allreduce_(coll_user_, v + off, m) != 0) return -1;
    }
    return 0;
  };
  // own-pose blocks of bs doubles -> every group; then the neighbour rows of v[a] (behind its n0 own rows) are filled
  auto share = [&](std::vector<std::vector<double>> &v, int bs) -> int {
    std::vector<double> glob((size_t)NP * bs, 0.0);
    for (int a = 0; a < N; a++)
      for (int k = 0; k < info_[a].n[0]; k++) std::copy(&v[a][(size_t)k * bs], &v[a][(size_t)(k + 1) * bs], &glob[(size_t)gid(a, k) * bs]);
    if (allsum(glob.data(), glob.size()) != 0) return -1;
    for (int a = 0; a < N; a++) {
      const int n0 = info_[a].n[0], n1 = info_[a].n[1];
      v[a].resize((size_t)(n0 + n1) * bs);
      for (int k = 0; k < n1; k++) {
        const size_t g = (size_t)key_gid(info_[a].nbr_key[k]) * bs;
        std::copy(&glob[g], &glob[g + bs], &v[a][(size_t)(n0 + k) * bs]);
      }
    }
    return 0;
  };
  // one block of len doubles per NODE of the graph (by node id): the owners fill theirs, everybody gets all of them
  auto share_nodes = [&](std::vector<std::vector<double>> &by_node, int len) -> int {
    std::vector<double> flat((size_t)Nn * len, 0.0);
    for (int a = 0; a < N; a++) std::copy(by_node[nodes_[a]].begin(), by_node[nodes_[a]].begin() + len, &flat[(size_t)nodes_[a] * len]);
    if (allsum(flat.data(), flat.size()) != 0) return -1;
    for (int b = 0; b < Nn; b++) by_node[b].assign(&flat[(size_t)b * len], &flat[(size_t)(b + 1) * len]);
    return 0;
  };
  // n_communicate (DChordalReduced.h:24-51): the neighbours' own blocks into Xk
  auto n_communicate = [&](std::vector<ReducedNode> &st) -> int {
    if (st.empty()) return 0;
    const int len = st[0].p * d;
    std::vector<double> flat((size_t)Nn * len, 0.0);
    for (int a = 0; a < N; a++) std::copy(st[a].Xk.begin(), st[a].Xk.begin() + len, &flat[(size_t)nodes_[a] * len]);
    if (allsum(flat.data(), flat.size()) != 0) return -1;
    for (auto &sn : st)
      for (const auto &bi : sn.n_index) {
        if (bi.first == sn.a) continue;
        std::copy(&flat[(size_t)bi.first * len], &flat[(size_t)(bi.first + 1) * len], sn.Xk.begin() + (size_t)bi.second * len);
      }
    return 0;
  };
  auto sum_groups = [&](double v) -> double {
    if (allsum(&v, 1) != 0) return NAN;
    return v;
  };
  // ---- stage 0: local solutions, X_a = [t (n0 x d) ; Y blocks], then the gauge of :156-157 (first rotation = I)
  std::vector<double> Xl;
  if (!Xlocal) {
    Graph gi;
    gi.d = d;
    gi.num_poses = NP;
    for (int a = 0; a < N; a++)
      for (Measurement m : info_[a].intra) {
        m.ipose = g_index_[a].at(m.ipose);
        m.jpose = g_index_[a].at(m.jpose);
        m.inode = m.jnode = 0;
        gi.all.push_back(m);
      }
    if (partition(gi, Nn) != 0) return -1;
    Xl.assign((size_t)(d + 1) * NP * d, 0.0);
    const int ldx = (d + 1) * NP;
    for (int a = 0; a < N; a++) {   // chordal initialisation of the node's own subgraph (local pose ids)
      Graph ga;
      ga.d = d;
      ga.num_poses = info_[a].n[0];
      for (Measurement m : info_[a].intra) {
        m.ipose = info_[a].tail(m);
        m.jpose = info_[a].head(m);
        m.inode = m.jnode = 0;
        ga.all.push_back(m);
      }
      const int n0 = ga.num_poses, lda = (d + 1) * n0;
      std::vector<double> Xa((size_t)lda * d);
      if (ga.all.empty() || chordal_initialization(ga, Xa.data(), lda) != 0) {
        fprintf(stderr, "[dpgo_amd] ERROR: local initialisation of node %d failed (are its poses connected by intra-node edges?).\n", a);
        return -1;
      }
      for (int k = 0; k < n0; k++)
        for (int c = 0; c < d; c++) {
          Xl[(size_t)c * ldx + gid(a, k)] = Xa[(size_t)c * lda + k];
          for (int r = 0; r < d; r++) Xl[(size_t)c * ldx + NP + gid(a, k) * d + r] = Xa[(size_t)c * lda + n0 + k * d + r];
        }
    }
    Options lo = opt_;
    lo.scheme = 0;            // MM-PGO
    lo.loss = 0;
    lo.accepted_delta = 0.0;  // refinement on in every iteration
    lo.preconditioner = 3;
    std::unique_ptr<Group> loc(new Group(gi, nodes_, lo, device_));
    std::vector<int> every(N);   // (update / iterate take LOCAL indices)
    for (int a = 0; a < N; a++) every[a] = a;
    if (!loc->ok() || loc->initialize_global(Xl.data(), ldx) != 0 || loc->update(every) != 0) return -1;
    for (int it = 0; it < o.local_iters; it++)
      if (loc->iterate(every) != 0 || loc->update(every) != 0) return -1;
    if (loc->scatter_global(Xl.data(), ldx) != 0) return -1;
    Xlocal = Xl.data();
    ldl = ldx;
  }
  clk.lap("dist-init: stage 0 (local solutions)");
  // xs[a]: (n0 + n1) poses, each [t (d) | Y (d x d)] -- own poses in the node's gauge, neighbours filled by communicate
  std::vector<std::vector<double>> xt(N), xY(N);
  auto fill_neighbours = [&]() -> int {   // DPGO::communicate (DPGO_utils.h:397-453)
    return (share(xt, d) != 0 || share(xY, d * d) != 0) ? -1 : 0;
  };
  for (int a = 0; a < N; a++) {
    const int n0 = info_[a].n[0], n1 = info_[a].n[1];
    xt[a].assign((size_t)(n0 + n1) * d, 0.0);
    xY[a].assign((size_t)(n0 + n1) * d * d, 0.0);
    double Y0[9], Yi[9], t[3];
    for (int r = 0; r < d; r++)
      for (int c = 0; c < d; c++) Y0[r * d + c] = Xlocal[(size_t)c * ldl + NP + gid(a, 0) * d + r];
    for (int k = 0; k < n0; k++) {
      for (int r = 0; r < d; r++)
        for (int c = 0; c < d; c++) Yi[r * d + c] = Xlocal[(size_t)c * ldl + NP + gid(a, k) * d + r];
      for (int c = 0; c < d; c++) t[c] = Xlocal[(size_t)c * ldl + gid(a, k)];
      mul_nn(d, Yi, Y0, &xY[a][(size_t)k * d * d]);                       // Y_i Y_0  (xhat^T xhat[:, n:n+d])
      for (int c = 0; c < d; c++) {
        double acc = 0;
        for (int q = 0; q < d; q++) acc += t[q] * Y0[q * d + c];
        xt[a][(size_t)k * d + c] = acc;
      }
    }
  }
  clk.lap("dist-init:   local gauges");
  if (fill_neighbours() != 0) return -1;
  clk.lap("dist-init:   neighbours' poses (shared)");
  std::vector<std::map<int, int>> nidx(N);
  for (int a = 0; a < N; a++) nidx[a] = make_n_index(info_[a]);
  auto sample = [&](double v) { if (objectives) objectives->push_back(v); };
  // ---- stage 1: reduced rotations (:160-225)
  std::vector<ReducedNode> rr(N);
  for (int a = 0; a < N; a++) {
    ReducedNode &s = rr[a];
    s.a = nodes_[a]; s.d = d; s.p = d; s.n_index = nidx[a]; s.nn = (int)nidx[a].size() - 1;
    const int rows = (s.nn + 1) * d;
    s.M = (int)info_[a].inter.size();
    std::vector<double> G((size_t)d * d, 0.0);
    s.S.assign((size_t)d * rows, 0.0);
    s.g.assign((size_t)d * d, 0.0);
    s.Bm.assign((size_t)s.M * d * rows, 0.0);
    s.b.assign((size_t)s.M * d * d, 0.0);
    int e = 0;
    for (const auto &m : info_[a].inter) {
      const int i = info_[a].tail(m), j = info_[a].head(m);
      const bool tail_local = m.inode == nodes_[a];
      double tmp[9], nR[9];
      mul_tn(d, &xY[a][(size_t)i * d * d], m.R, tmp);      // Y_i^T R_e
      mul_nn(d, tmp, &xY[a][(size_t)j * d * d], nR);       // ... Y_j          (:255-258)
      const int ni0 = tail_local ? 0 : nidx[a].at(m.inode), ni1 = tail_local ? nidx[a].at(m.jnode) : 0;
      const double sk = std::sqrt(m.kappa);
      for (int r = 0; r < d; r++) {
        for (int c = 0; c < d; c++) s.Bm[(size_t)(e * d + r) * rows + ni0 * d + c] += sk * nR[c * d + r];
        s.Bm[(size_t)(e * d + r) * rows + ni1 * d + r] -= sk;
        G[r * d + r] += 2 * m.kappa;
        s.S[(size_t)r * rows + r] -= m.kappa;
        for (int c = 0; c < d; c++)
          s.S[(size_t)r * rows + (tail_local ? ni1 : ni0) * d + c] -= m.kappa * (tail_local ? nR[r * d + c] : nR[c * d + r]);
      }
      e++;
    }
    for (int k = 0; k < d; k++) { G[k * d + k] += o.reg_G; s.S[(size_t)k * rows + k] -= o.reg_G; }
    s.Ginv.resize((size_t)d * d);
    inv_small(d, G.data(), s.Ginv.data());
    std::vector<double> R0((size_t)rows * d, 0.0);
    for (int k = 0; k <= s.nn; k++)
      for (int r = 0; r < d; r++) R0[(size_t)(k * d + r) * d + r] = 1.0;
    s.initialize(R0);
  }
  clk.lap("dist-init:   reduced rotation problems");
  for (int it = 0; it < o.iters[0]; it++) {
    if (it % 20 == 0) {
      double f = 0;
      for (auto &s : rr) f += s.objective();
      sample(0.5 * sum_groups(f));
    }
    for (int a = 0; a < N; a++)
      if (nodes_[a] != 0) { rr[a].update(); rr[a].iterate(); }   // node 0 keeps its gauge (:160-225)
    if (n_communicate(rr) != 0) return -1;
  }
  std::vector<std::vector<double>> rots_n(Nn, std::vector<double>((size_t)d * d, 0.0));   // by node id
  for (int a = 0; a < N; a++) {
    rots_n[nodes_[a]].assign(rr[a].Xk.begin(), rr[a].Xk.begin() + d * d);
    project_to_SOd_host(d, rots_n[nodes_[a]].data());
  }
  if (share_nodes(rots_n, d * d) != 0) return -1;
  clk.lap("dist-init: stage 1 (reduced rotations, host)");
  // ---- stage 2: rotations on the device (:230-304)
  std::vector<std::vector<double>> rots(N);
  auto halo = [&](std::vector<std::vector<double>> &v, int bs) -> int { return share(v, bs); };   // DChordal::communicate (DChordal_utils.h:196-240)
  for (int a = 0; a < N; a++) {
    const int n0 = info_[a].n[0];
    rots[a].assign((size_t)n0 * d * d, 0.0);
    for (int k = 0; k < n0; k++) mul_nn(d, &xY[a][(size_t)k * d * d], rots_n[nodes_[a]].data(), &rots[a][(size_t)k * d * d]);
  }
  if (halo(rots, d * d) != 0) return -1;
  if (chordal_setup(0, o.reg_G, {}) != 0 || chordal_initialize(rots) != 0) return -1;
  for (int it = 0; it < o.iters[1]; it++) {
    if (it % 20 == 0) sample(sum_groups(chordal_objective()));
    if (chordal_step() != 0) return -1;
  }
  if (chordal_get(rots) != 0) return -1;
  for (int a = 0; a < N; a++) {
    const int n0 = info_[a].n[0];
    for (int k = 0; k < n0; k++) project_to_SOd_host(d, &rots[a][(size_t)k * d * d]);   // per-block projection (:292-298)
    rots_n[nodes_[a]].assign(rots[a].begin(), rots[a].begin() + d * d);
    for (int k = 0; k < n0; k++) mul_nt(d, &rots[a][(size_t)k * d * d], rots_n[nodes_[a]].data(), &xY[a][(size_t)k * d * d]);   // back to the node's gauge
  }
  if (share_nodes(rots_n, d * d) != 0) return -1;
  clk.lap("dist-init: stage 2 (rotations, device)");
  // ---- stage 3: reduced translations (:311-359)
  {
    // recover_translations (DChordalReducedProblem.h:251-261) of every node: t = -L^-1 (P R), L = intra tau-Laplacian + 100
    // at (0,0).  The nodes' systems are the diagonal blocks of one matrix: one ordering (components in parallel), one
    // numeric factorisation, one solve.
    CsrMatrix Lm;
    Lm.ptr.push_back(0);
    std::vector<double> rhs((size_t)P0_ * d, 0.0);
    for (int a = 0; a < N; a++) {
      const DataInfo &info = info_[a];
      const int n0 = info.n[0], off = own_off_[a];
      std::vector<std::map<int, double>> rows(n0);
      for (const auto &m : info.intra) {
        const int i = info.tail(m), j = info.head(m);
        rows[i][i] += m.tau; rows[j][j] += m.tau; rows[i][j] -= m.tau; rows[j][i] -= m.tau;
        for (int c = 0; c < d; c++) {
          double acc = 0;
          for (int k = 0; k < d; k++) acc += m.t[k] * xY[a][(size_t)i * d * d + k * d + c];
          rhs[(size_t)(off + i) * d + c] += m.tau * acc;
          rhs[(size_t)(off + j) * d + c] -= m.tau * acc;
        }
      }
      rows[0][0] += 100.0;
      for (int r = 0; r < n0; r++) {
        for (const auto &cv : rows[r]) { Lm.col.push_back(off + cv.first); Lm.val.push_back(cv.second); }
        Lm.ptr.push_back((int)Lm.col.size());
      }
    }
    Lm.n = (int)Lm.ptr.size() - 1;
    SpdFactor F;
    if (spd_factor(Lm, F, 64, 1) != 0) return -1;
    spd_solve_host(F, rhs.data(), d);
    for (int a = 0; a < N; a++) {
      const int n0 = info_[a].n[0], off = own_off_[a];
      for (int k = 0; k < n0; k++)
        for (int c = 0; c < d; c++) xt[a][(size_t)k * d + c] = -(rhs[(size_t)(off + k) * d + c] - rhs[(size_t)off * d + c]);
    }
  }
  if (fill_neighbours() != 0) return -1;
  std::vector<std::vector<double>> nRs(N);
  for (int a = 0; a < N; a++) {   // DChordal::n_communicate(problems_red_R, rots_n)  (DChordal_utils.h:148-190)
    nRs[a].assign((size_t)nidx[a].size() * d * d, 0.0);
    for (const auto &bi : nidx[a]) std::copy(rots_n[bi.first].begin(), rots_n[bi.first].begin() + d * d, &nRs[a][(size_t)bi.second * d * d]);
  }
  std::vector<ReducedNode> rt(N);
  for (int a = 0; a < N; a++) {
    ReducedNode &s = rt[a];
    s.a = nodes_[a]; s.d = d; s.p = 1; s.n_index = nidx[a]; s.nn = (int)nidx[a].size() - 1;
    const int rows = s.nn + 1;
    s.M = (int)info_[a].inter.size();
    double G = 0;
    s.S.assign(rows, 0.0);
    s.g.assign(d, 0.0);
    s.Bm.assign((size_t)s.M * rows, 0.0);
    s.b.assign((size_t)s.M * d, 0.0);
    int e = 0;
    for (const auto &m : info_[a].inter) {
      const int i = info_[a].tail(m), j = info_[a].head(m);
      const bool tail_local = m.inode == nodes_[a];
      const int ni0 = tail_local ? 0 : nidx[a].at(m.inode), ni1 = tail_local ? nidx[a].at(m.jnode) : 0;
      double ti[3], nt[3];
      for (int c = 0; c < d; c++) {
        double acc = xt[a][(size_t)i * d + c];
        for (int k = 0; k < d; k++) acc += xY[a][(size_t)i * d * d + k * d + c] * m.t[k];
        ti[c] = acc;
      }
      const double *tj = &xt[a][(size_t)j * d], *nRi = &nRs[a][(size_t)ni0 * d * d], *nRj = &nRs[a][(size_t)ni1 * d * d];
      for (int c = 0; c < d; c++) {
        double acc = 0;
        for (int k = 0; k < d; k++) acc += nRi[k * d + c] * ti[k] - nRj[k * d + c] * tj[k];
        nt[c] = acc;
      }
      const double st = std::sqrt(m.tau), sg = tail_local ? 1.0 : -1.0;
      s.Bm[(size_t)e * rows + ni0] += sg * st;
      s.Bm[(size_t)e * rows + ni1] -= sg * st;
      G += 2 * m.tau;
      s.S[0] -= m.tau;
      s.S[tail_local ? ni1 : ni0] -= m.tau;
      for (int c = 0; c < d; c++) { s.b[(size_t)e * d + c] = sg * st * nt[c]; s.g[c] += sg * m.tau * nt[c]; }
      e++;
    }
    G += o.reg_G;
    s.S[0] -= o.reg_G;
    s.Ginv.assign(1, 1.0 / G);
    s.initialize(std::vector<double>((size_t)rows * d, 0.0));
  }
  for (int it = 0; it < o.iters[2]; it++) {
    if (it % 20 == 0) {
      double f = 0;
      for (auto &s : rt) f += s.objective();
      sample(0.5 * sum_groups(f));
    }
    for (int a = 0; a < N; a++)
      if (nodes_[a] != 0) { rt[a].update(); rt[a].iterate(); }
    if (n_communicate(rt) != 0) return -1;
  }
  clk.lap("dist-init: stage 3 (reduced translations, host)");
  // ---- stage 4: translations on the device (:365-407)
  std::vector<std::vector<double>> ts(N);
  for (int a = 0; a < N; a++) {
    const int n0 = info_[a].n[0];
    ts[a].assign((size_t)n0 * d, 0.0);
    for (int k = 0; k < n0; k++)
      for (int c = 0; c < d; c++) {
        double acc = rt[a].Xk[c];
        for (int q = 0; q < d; q++) acc += xt[a][(size_t)k * d + q] * rots_n[nodes_[a]][q * d + c];
        ts[a][(size_t)k * d + c] = acc;
      }
    rots[a].resize((size_t)n0 * d * d);
  }
  if (halo(rots, d * d) != 0 || halo(ts, d) != 0) return -1;
  if (chordal_setup(1, o.reg_G, rots) != 0 || chordal_initialize(ts) != 0) return -1;
  for (int it = 0; it < o.iters[3]; it++) {
    if (it % 20 == 0) sample(sum_groups(chordal_objective()));
    if (chordal_step() != 0) return -1;
  }
  if (chordal_get(ts) != 0) return -1;
  chordal_release();
  // Xk[alpha] = [t ; R] of every node into the global X (:409-415, then :466-475)
  for (int a = 0; a < N; a++)
    for (int k = 0; k < info_[a].n[0]; k++) {
      const int gp = gid(a, k);
      for (int c = 0; c < d; c++) {
        X[(size_t)c * ld + gp] = ts[a][(size_t)k * d + c];
        for (int r = 0; r < d; r++) X[(size_t)c * ld + NP + gp * d + r] = rots[a][(size_t)k * d * d + r * d + c];
      }
    }
  if (spread) {   // every group wrote its own poses: the sum over the groups is the whole initial guess
    for (int c = 0; c < d; c++) {
      double *col = X + (size_t)c * ld;
      // (rows of poses hosted elsewhere must be zero before the sum)
      std::vector<char> mine((size_t)(d + 1) * NP, 0);
      for (int a = 0; a < N; a++)
        for (int k = 0; k < info_[a].n[0]; k++) {
          const int gp = gid(a, k);
          mine[gp] = 1;
          for (int r = 0; r < d; r++) mine[(size_t)NP + gp * d + r] = 1;
        }
      for (size_t i = 0; i < mine.size(); i++)
        if (!mine[i]) col[i] = 0.0;
      if (allsum(col, (size_t)(d + 1) * NP) != 0) return -1;
    }
  }
  return 0;
}

}  // namespace dpgo
