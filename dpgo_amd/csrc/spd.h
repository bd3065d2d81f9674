// Sparse SPD direct solver: nested-dissection multifrontal Cholesky.  Separators: minimum vertex covers of BFS-level
// and spectral (Fiedler) cuts, see Dissector in spd.cpp.
//
// Replaces the reference's CHOLMOD factorisations `L_` (of G_tt) and
// `reg_Chol_precon_` (of G_RR + lambda I): C++/DPGO/src/DPGOProblem.cpp:93,119;
// solves at DPGOProblem.h:291, DPGOProblem.cpp:140,568,592.
//
// MI355X-first design: the factorisation happens once on the host (setup, as in
// the reference); what runs per MM iteration is the SOLVE, and a level-scheduled
// sparse triangular solve is latency-poison on a GPU.  So every front s stores
//     W_s = [ L11^-1 ; -L21 L11^-1 ]        ((w+u) x w, dense)
// which turns both sweeps into dense mat-vecs with no triangular dependency
// inside a front:
//     forward :  [y_s ; dupd] = W_s f_s          (f_s = rhs + children's updates)
//     backward:  x_s = W_s^T [y_s ; x_upd]
// Fronts of the same tree height (forward) / depth (backward) are independent and
// run in one launch; each front's update vector is pulled (not pushed) by its
// parent through precomputed gather lists, so the solve is deterministic and
// atomic-free.  W is kept in both orientations so both sweeps read coalesced.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

namespace dpgo {

struct SpdNumericCtx;

struct CsrMatrix {
  int n = 0;
  std::vector<int> ptr, col;
  std::vector<double> val;
};

struct SpdFactorData {
  int n = 0;
  int nfronts = 0;
  std::vector<int> w, u;                 // pivots / update rows per front
  std::vector<int> piv_ptr, piv_idx;     // CSR: matrix indices of the pivots
  std::vector<int> upd_ptr, upd_idx;     // CSR: matrix indices of the update rows
  std::vector<int64_t> w_off, wt_off;    // offset of W_s in W / of WT_s in WT
  std::vector<int> ldw, ldm;             // padded leading dimensions of W_s (>= w) and WT_s (>= w+u)
  int64_t entries = 0;                   // sum (w+u) w: factor entries without padding
  std::vector<double> W;                 // (w+u) x ldw row-major per front
  std::vector<double> WT;                // w x ldm row-major per front
  std::vector<int> parent, height, depth;
  std::vector<int> pos_off;              // first "front position" of each front (w+u positions each)
  std::vector<int> ubuf_off;             // first row of each front's update vector in the update buffer
  std::vector<int> asm_ptr, asm_src;     // per front position: rows of the update buffer to add
  std::vector<std::vector<int>> by_height, by_depth;
  std::vector<std::vector<int>> children;   // elimination tree, kept for spd_refactor
  int total_pos = 0, total_upd = 0, max_front = 0;
  // keep_device: the numeric phase leaves W / WT on the GPU (dev_W / dev_WT, same per-front layout as W / WT, owned by
  // the factor until spd_release_device) and does not fill the host copies -- for callers that only solve on the device
  // smallest / largest pivot d_kk of the factorisation (device numeric phase; 0 / 0 when not recorded): their ratio is a
  // lower bound of the condition number -- W_s holds the explicit L11^-1, so a huge ratio costs digits in every solve
  double pivot_min = 0.0, pivot_max = 0.0;
  bool keep_device = false;
  // keep_numeric (with keep_device): the device state of the numeric phase stays with the factor (`numeric`), dev_W / dev_WT
  // are borrowed from it, and spd_refactor_device() re-factors from values that are already on the GPU
  bool keep_numeric = false, dev_borrowed = false;
  struct SpdNumericCtx *numeric = nullptr;
  int leaf = 32, collapse = 0, block = 1;   // the parameters this factor was built with (spd_refactor's host path repeats them)
  double *dev_W = nullptr, *dev_WT = nullptr;
  int64_t nnz() const { return entries; }
};
// The factor OWNS `numeric` and (unless borrowed from it) dev_W / dev_WT -- released by spd_release_numeric /
// spd_release_device, which the holder calls -- so a copy would free them twice: factors move, they are never copied.
struct SpdFactor : SpdFactorData {
  SpdFactor() = default;
  SpdFactor(const SpdFactor &) = delete;
  SpdFactor &operator=(const SpdFactor &) = delete;
  SpdFactor(SpdFactor &&o) noexcept : SpdFactorData(std::move(static_cast<SpdFactorData &>(o))) { o.disown(); }
  SpdFactor &operator=(SpdFactor &&o) noexcept {
    if (this != &o) {
      static_cast<SpdFactorData &>(*this) = std::move(static_cast<SpdFactorData &>(o));
      o.disown();
    }
    return *this;
  }

 private:
  void disown() { numeric = nullptr; dev_W = dev_WT = nullptr; dev_borrowed = false; }
};

// Factor A (symmetric positive definite, full pattern in CSR).  leaf = max
// vertices of a nested-dissection leaf.  Returns 0, or -1 if a pivot is not positive.
// collapse = number of nested-dissection levels merged into one front (1 = plain binary tree,
// 0 = choose 1..5 from a latency + bandwidth model of the device solve).
// block > 1: consecutive groups of `block` unknowns share their neighbours (the d rotation rows of a pose); the ordering
// is computed on the quotient graph
int spd_factor(const CsrMatrix &A, SpdFactor &F, int leaf = 32, int collapse = 0, int block = 1, bool keep_device = false);
// frees dev_W / dev_WT (no-op when there are none)
void spd_release_device(SpdFactor &F);

// New values, same pattern (a Dynamic rescale changes the diagonal of G_tt): the numeric phase only, on the GPU
// (spd_dev.hip); without a GPU the whole factorisation is redone.  F must come from spd_factor of the same pattern.
int spd_refactor(const CsrMatrix &A, SpdFactor &F);

// With keep_numeric: the device copy of A's values (CSR order of the matrix that was factored), the numeric phase from
// them, and the release of the kept state (a factor that owns one must not be copied)
double *spd_numeric_values(SpdFactor &F);
// stream (a hipStream_t, nullptr: the factorisation's own): where the kernels run; defer: return as soon as they are
// enqueued -- the caller queues what depends on the new factor behind them on the same stream and calls
// spd_refactor_finish, which waits and returns -1 on a non-positive pivot
int spd_refactor_device(SpdFactor &F, void *stream = nullptr, bool defer = false);
int spd_refactor_finish(SpdFactor &F, bool wait = true);   // wait = false: the stream is known to have passed the factorisation
void spd_release_numeric(SpdFactor &F);

// Host solve (setup paths and tests): X (n x ncols, row-major) <- A^-1 X.
void spd_solve_host(const SpdFactor &F, double *X, int ncols);

}  // namespace dpgo
