// Host-side assembly of one node's surrogate operators in pose-contiguous
// block form.
//
// The reference assembles Eigen RowMajor sparse matrices over the row order
// [t^a; R^a; t^nbr; R^nbr] from std::list triplets
// (C++/DPGO/src/DPGO_utils.cpp:1398-2288 trivial loss, :2290-2967 Static robust).
// Here every operator is a block-sparse matrix over POSES: block (p,q) is the
// (d+1)x(d+1) coupling between pose records z_p = [x_p; Y_p] ((d+1) x d, row 0
// the translation, rows 1..d the rows of R_p^T), laid out contiguously so a
// thread reads whole blocks / records with 16-byte loads.  Pose ids are local to
// the node: own pose k -> k, neighbour pose k -> n0 + k.
#pragma once
#include <vector>

#include "graph.h"
#include "spd.h"

namespace dpgo {

struct BsrMatrix {
  int B = 0;          // block dimension d+1
  int nrows = 0;      // block rows
  int ncols = 0;      // block columns
  std::vector<int> ptr, col;
  std::vector<double> val;   // B*B row-major per block
};

struct NodeOperators {
  int d = 0, n0 = 0, n1 = 0;
  bool trivial = true;
  BsrMatrix G;        // own x own           (all losses)
  BsrMatrix S;        // own x (own+nbr)     (trivial)
  BsrMatrix P, P0;    // (own+nbr)^2         (trivial)
  BsrMatrix Q;        // (own+nbr)^2: trivial -> -1/2 E+ - xi ; robust -> block diagonal 2 bd(E) + 2 xi
  std::vector<double> D;      // own block diagonal, B*B per pose
  std::vector<double> Hd;     // diagonal blocks of the proximal majoriser H (what T, N, V are made of), B*B per pose
  std::vector<double> Tinv;   // T_i = 1 / h_tt                         (n0)
  std::vector<double> N;      // N_i = T_i h_tR                         (n0 x d)
  std::vector<double> V;      // V_i = H_RR - h_tR^T T_i h_tR           (n0 x d x d)
  CsrMatrix Gtt;              // n0 x n0 scalar
  CsrMatrix GRR;              // d n0 x d n0 scalar
};

// xi = options.regularizer.  trivial selects simplify_quadratic_data_matrix,
// otherwise simplify_regular_data_matrix (Static rescale).
// scale (optional, one number per inter-node edge in the order of info.inter): Rescale::Dynamic.
int assemble_node(const DataInfo &info, double xi, bool trivial, NodeOperators &ops, const double *scale = nullptr);

}  // namespace dpgo
