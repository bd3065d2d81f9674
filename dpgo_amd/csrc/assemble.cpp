#include "assemble.h"

#include <algorithm>
#include <cstring>
#include <map>

namespace dpgo {
namespace {

// Accumulator for a block-sparse matrix: one ordered map of blocks per row.
struct BlockAcc {
  int B, nrows, ncols;
  std::vector<std::map<int, std::vector<double>>> rows;
  BlockAcc(int B_, int nr, int nc) : B(B_), nrows(nr), ncols(nc), rows(nr) {}
  double *at(int r, int c) {
    auto &v = rows[r][c];
    if (v.empty()) v.assign((size_t)B * B, 0.0);
    return v.data();
  }
  void add(int r, int c, const double *blk, double scale) {
    if (r < 0 || r >= nrows) return;   // row outside this operator (e.g. neighbour row of an own-row matrix)
    double *dst = at(r, c);
    for (int k = 0; k < B * B; k++) dst[k] += scale * blk[k];
  }
  void add_diag(int r, double v, int from = 0) {
    double *dst = at(r, r);
    for (int k = from; k < B; k++) dst[k * B + k] += v;
  }
  void finish(BsrMatrix &M) const {
    M.B = B; M.nrows = nrows; M.ncols = ncols;
    M.ptr.assign(nrows + 1, 0);
    for (int r = 0; r < nrows; r++) M.ptr[r + 1] = M.ptr[r] + (int)rows[r].size();
    M.col.resize(M.ptr[nrows]);
    M.val.resize((size_t)M.ptr[nrows] * B * B);
    for (int r = 0; r < nrows; r++) {
      int k = M.ptr[r];
      for (const auto &kv : rows[r]) {
        M.col[k] = kv.first;
        std::memcpy(&M.val[(size_t)k * B * B], kv.second.data(), sizeof(double) * B * B);
        k++;
      }
    }
  }
};

// The four (d+1)x(d+1) blocks of the edge Hessian E_e in the slot order
// [t, R rows]: AA (tail,tail), AI (tail,head), IA (head,tail), II (head,head).
// Same entries as the triplets of DPGO_utils.cpp:1541-1641.
struct EdgeBlocks {
  double AA[16], AI[16], IA[16], II[16];
};

void edge_blocks(const Measurement &m, int d, EdgeBlocks &E) {
  const int B = d + 1;
  std::memset(&E, 0, sizeof(E));
  const double tau = m.tau, kap = m.kappa;
  E.AA[0] = tau;
  E.AI[0] = -tau;
  E.IA[0] = -tau;
  E.II[0] = tau;
  for (int k = 0; k < d; k++) {
    E.AA[0 * B + 1 + k] = tau * m.t[k];
    E.AA[(1 + k) * B + 0] = tau * m.t[k];
    E.IA[0 * B + 1 + k] = -tau * m.t[k];        // (t_j, R_i)
    E.AI[(1 + k) * B + 0] = -tau * m.t[k];      // (R_i, t_j)
    E.AA[(1 + k) * B + 1 + k] += kap;
    E.II[(1 + k) * B + 1 + k] += kap;
  }
  for (int r = 0; r < d; r++)
    for (int c = 0; c < d; c++) {
      E.AA[(1 + r) * B + 1 + c] += tau * m.t[r] * m.t[c];
      E.AI[(1 + r) * B + 1 + c] = -kap * m.R[r * d + c];
      E.IA[(1 + r) * B + 1 + c] = -kap * m.R[c * d + r];
    }
}

void scalar_csr(const BsrMatrix &G, int n0, int d, bool rot, CsrMatrix &out) {
  // rot=false: the (t,t) entry of every block -> n0 x n0; rot=true: the (R,R) d x d part -> d n0 x d n0
  const int B = d + 1, dof = rot ? d : 1;
  out.n = n0 * dof;
  out.ptr.assign(out.n + 1, 0);
  out.col.clear();
  out.val.clear();
  for (int p = 0; p < n0; p++)
    for (int r = 0; r < dof; r++) {
      for (int k = G.ptr[p]; k < G.ptr[p + 1]; k++) {
        const double *blk = &G.val[(size_t)k * B * B];
        for (int c = 0; c < dof; c++) {
          double v = rot ? blk[(1 + r) * B + 1 + c] : blk[0];
          if (rot && v == 0.0 && !(G.col[k] == p && r == c)) continue;
          out.col.push_back(G.col[k] * dof + c);
          out.val.push_back(v);
        }
      }
      out.ptr[p * dof + r + 1] = (int)out.col.size();
    }
}

}  // namespace

int assemble_node(const DataInfo &info, double xi, bool trivial, NodeOperators &ops, const double *scale) {
  // scale != nullptr: Rescale::Dynamic (DPGO_utils.cpp:2969-3903 + DPGOProblem::update_quadratic_mat,
  // DPGOProblem.cpp:751-840): the contribution of inter-node edge e to G, D, Q and to the proximal majoriser H
  // (the columns of the reference's E_ / F_, :3518-3585) is multiplied by scale[e], and the majoriser is
  // regularised by 0.5 xi instead of 1.5 xi (:3621, :3634 against :2222-2241).
  const bool dynamic = scale != nullptr && !trivial;
  const int d = info.d, B = d + 1, n0 = info.n[0], n1 = info.n[1], nz = n0 + n1;
  ops = NodeOperators();
  ops.d = d; ops.n0 = n0; ops.n1 = n1; ops.trivial = trivial;
  BlockAcc G(B, n0, n0), H(B, n0, n0), Dd(B, n0, n0), Q(B, nz, nz);
  BlockAcc S(B, n0, nz), P(B, nz, nz), P0(B, nz, nz);
  EdgeBlocks E;
  for (const auto &m : info.intra) {
    const int p = info.tail(m), q = info.head(m);
    edge_blocks(m, d, E);
    // G += E (DPGO_utils.cpp:1541-1641); H += 2 bd(E) (:1679-1738)
    G.add(p, p, E.AA, 1); G.add(p, q, E.AI, 1); G.add(q, p, E.IA, 1); G.add(q, q, E.II, 1);
    H.add(p, p, E.AA, 2); H.add(q, q, E.II, 2);
    if (trivial) {  // P -= E (:1552-1640)
      P.add(p, p, E.AA, -1); P.add(p, q, E.AI, -1); P.add(q, p, E.IA, -1); P.add(q, q, E.II, -1);
    }
  }
  int e_inter = 0;
  for (const auto &m : info.inter) {
    const int p = info.tail(m), q = info.head(m);
    const bool tail_own = (m.inode == info.node);
    edge_blocks(m, d, E);
    const double w2 = 2.0 * (dynamic ? scale[e_inter] : 1.0);
    e_inter++;
    // own endpoint gets 2 bd(E) in G, D, H (:1964-2028, :2097-2127 / :2748-2865)
    if (tail_own) { G.add(p, p, E.AA, w2); Dd.add(p, p, E.AA, w2); H.add(p, p, E.AA, w2); }
    else          { G.add(q, q, E.II, w2); Dd.add(q, q, E.II, w2); H.add(q, q, E.II, w2); }
    if (trivial) {
      // Q = -1/2 E+ (:1864-1962), P0 = +1/2 E+ (:1872-1961), E+ = bd(E) - od(E)
      Q.add(p, p, E.AA, -0.5); Q.add(q, q, E.II, -0.5); Q.add(p, q, E.AI, 0.5); Q.add(q, p, E.IA, 0.5);
      P0.add(p, p, E.AA, 0.5); P0.add(q, q, E.II, 0.5); P0.add(p, q, E.AI, -0.5); P0.add(q, p, E.IA, -0.5);
      // P -= od(E) (:1869-1948)
      P.add(p, q, E.AI, -1); P.add(q, p, E.IA, -1);
      // S = -E+[own endpoint rows, :] (:1971-2035, :2105-2134)
      if (tail_own) { S.add(p, p, E.AA, -1); S.add(p, q, E.AI, 1); }
      else          { S.add(q, q, E.II, -1); S.add(q, p, E.IA, 1); }
    } else {
      // robust Q = 2 bd(E) on both endpoints (:2711-2746)
      Q.add(p, p, E.AA, w2); Q.add(q, q, E.II, w2);
    }
  }
  for (int i = 0; i < n0; i++) {   // xi terms (:2212-2243 / :2906-2927)
    G.add_diag(i, xi); Dd.add_diag(i, xi); H.add_diag(i, (dynamic ? 0.5 : 1.5) * xi);
    if (trivial) { S.add_diag(i, -xi); Q.add_diag(i, -xi); P.add_diag(i, xi); P0.add_diag(i, xi); }
    else Q.add_diag(i, 2.0 * xi);
  }
  G.finish(ops.G);
  Q.finish(ops.Q);
  if (trivial) { S.finish(ops.S); P.finish(ops.P); P0.finish(ops.P0); }
  ops.D.assign((size_t)n0 * B * B, 0.0);
  ops.Hd.assign((size_t)n0 * B * B, 0.0);
  ops.Tinv.assign(n0, 0.0);
  ops.N.assign((size_t)n0 * d, 0.0);
  ops.V.assign((size_t)n0 * d * d, 0.0);
  for (int i = 0; i < n0; i++) {
    std::memcpy(&ops.D[(size_t)i * B * B], Dd.at(i, i), sizeof(double) * B * B);
    const double *h = H.at(i, i);
    std::memcpy(&ops.Hd[(size_t)i * B * B], h, sizeof(double) * B * B);
    // T = diag(H_tt)^-1, N = T H_tR, V = H_RR - H_Rt T H_tR (:2280-2282 / :2958-2964)
    const double T = 1.0 / h[0];
    ops.Tinv[i] = T;
    for (int k = 0; k < d; k++) ops.N[(size_t)i * d + k] = T * h[1 + k];
    for (int r = 0; r < d; r++)
      for (int c = 0; c < d; c++)
        ops.V[((size_t)i * d + r) * d + c] = h[(1 + r) * B + 1 + c] - h[(1 + r) * B] * (T * h[1 + c]);
  }
  scalar_csr(ops.G, n0, d, false, ops.Gtt);
  scalar_csr(ops.G, n0, d, true, ops.GRR);
  return 0;
}

}  // namespace dpgo
