// Launch wrappers for the hand-written gfx950 kernels (kernels.hip).
//
// Conventions shared by every kernel:
//  * A pose record z_p is (d+1) x d doubles, row-major: row 0 = translation
//    x_p, rows 1..d = rows of Y_p = R_p^T.  RS = (d+1)*d doubles (96 B for SE(3),
//    48 B for SE(2)), 16-byte aligned, poses contiguous.  The reference stores
//    the same numbers as rows {p, n + d p .. n + d p + d - 1} of a column-major
//    Eigen matrix (C++/DPGO/include/DPGO/DPGOProblem.h:167-171).
//  * One device hosts several nodes.  Rows are "unified": own poses of all local
//    nodes first (node by node), then all neighbour poses (node by node).
//  * Work is cut into segments of <= SEG_ROWS rows that never straddle a node;
//    one workgroup per segment.  Kernels that reduce write one partial per
//    (slot, segment); k_reduce sums a node's partials in fixed order
//    (deterministic, no atomics).
//  * every launch carries a NodeMask (bit a = local node a): workgroups of nodes outside it exit early; this
//    is how per-node branches of the AMM state machine run without splitting launches.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dpgo {

// 64 rows per segment = one wavefront per workgroup for the row kernels: a node of 12.5 k poses still
// yields ~200 workgroups, enough to spread over the 256 CUs when a GPU holds a single node.
constexpr int SEG_ROWS = 64;
// lanes that share one block row in k_bsr (lane j takes blocks j, j + BSR_LPR, ...); the block values are stored
// interleaved per round of BSR_LPR blocks (Group::upload_bsr)
#ifndef DPGO_BSR_LPR
#define DPGO_BSR_LPR 4   // (8 lanes per row measured 5 % slower at the headline size, equal at one node per GPU)
#endif
constexpr int BSR_LPR = DPGO_BSR_LPR;
constexpr int MAX_SLOTS = 32;   // per-node scalars one read-back can carry; slots 16.. hold the first CG step's sums (k_cg_scal_begin),
constexpr int UPD_SLOT0 = 24;   // slots 24.. the sums of update(): they wait there for the next refinement's k_cg_scal_begin to reduce them
constexpr int MAX_DOTS = 6;     // dot products per k_dots launch (it stores MAX_DOTS consecutive slots)

struct Seg {
  int begin, end, node, pad;
};

// Two coefficients per local node, passed BY VALUE as a kernel argument (no upload, no memory latency): the step
// lengths of the batched CG.  A group hosts at most MAX_LOCAL_NODES nodes.
constexpr int MAX_LOCAL_NODES = 64;
// which of the group's nodes a launch works on: bit a = local node a.  `v` travels by value like the coefficients;
// `p` (optional) points at a device-resident mask that the kernel ands in -- the truncated CG keeps its set of
// still-iterating nodes on the device (k_cg_scal), so the host need not read it back before the next launch
typedef unsigned long long NodeBits;
// A launch over the OWN segments of a few nodes only (the late steps of the truncated CG run on one to four of a group's
// nodes): with nlive > 0 the grid holds nlive * max(nseg) workgroups and workgroup b works on segment seg0[b % nlive] +
// b / nlive -- a node's segments then spread over all eight XCDs (round-robin dispatch), where the whole-group grid,
// which gives every XCD one contiguous eighth of the rows, would leave a single live node to one XCD; surplus workgroups
// land on idle_seg, a segment of a node outside v, and leave at once.  nlive = 0: the grid covers every segment.
constexpr int MAX_LIVE_SEGS = 8;
struct NodeMask {
  NodeBits v;
  const NodeBits *p;
  int nlive = 0, idle_seg = 0;
  int seg0[MAX_LIVE_SEGS] = {0, 0, 0, 0, 0, 0, 0, 0}, nseg[MAX_LIVE_SEGS] = {0, 0, 0, 0, 0, 0, 0, 0};
};
constexpr NodeMask ALL_NODES = {~0ull, nullptr};
struct NodeCoefs {
  double a[MAX_LOCAL_NODES], b[MAX_LOCAL_NODES];
};

// Per-node state of the Steihaug-Toint CG (IterativeSolvers.h:207-390), device resident: the scalar recurrences
// run in k_cg_scal, the vector kernels read their step lengths from here.
struct alignas(16) CgNode {
  double sk_M_pk, sk_M_2, pk_M_2, rv, Delta, Delta_2, target, h_M_norm;
  double c1, cr;        // s += c1 p, H s += c1 H p, r += cr H p   (cr = 0: the node stops after this step)
  double al, kap, be;   // alpha_k, kappa_k (kept for beta), beta_k
  int cg_it, max_it, live;
  int stop_ord;         // which scalar step ended the run: 2 j - 1 / 2 j for phase 0 / 1 of step j, 0: never started (CG_LIVE_ORD while live)
};
// start values of a CG run, by value (the host knows them from the read-back of the gradient norms)
struct CgStart {
  double rv[MAX_LOCAL_NODES], Delta[MAX_LOCAL_NODES], target[MAX_LOCAL_NODES];
};

struct BsrDev {
  int nrows = 0, nnzb = 0;
  const int *ptr = nullptr, *col = nullptr;
  const double *val = nullptr;
};

// Inter-node edges of the local nodes, B-form (residual) data.
// One incidence of an inter-node edge (edge e seen from one of its poses), everything k_inter needs in ONE 128-byte
// record: the other endpoint, the edge and the role (code = 2 e + (0 tail | 1 head)), the measurement.  The records
// of a row are consecutive (the order of inc): a row's chain is inc_ptr -> record -> the other pose, and an incidence
// costs one cache line instead of the six that tail/head, R, t, kappa, tau in separate arrays touch.
struct alignas(16) InterInc {
  int other, code;
  double tau, kappa, t[3], R[9];
  int osrc, pad;   // osrc: where the other pose's record is in the receive buffer of a lazy unpack (InterEdgesDev::recv), -1: in the record array
};
static_assert(sizeof(InterInc) == 128, "InterInc is loaded as eight 16-byte quads");
struct InterEdgesDev {
  const InterInc *rec = nullptr;                // per incidence (optional: k_inter; the arrays below serve k_cost)
  int m = 0, nrows_own = 0, nrows_all = 0;
  const int *tail = nullptr, *head = nullptr;   // unified pose ids
  const double *R = nullptr;                    // d*d row-major
  const double *t = nullptr;                    // d
  const double *kappa = nullptr, *tau = nullptr;
  const int *inc_ptr = nullptr;                 // per unified row
  const int *inc = nullptr;                     // edge*2 + (0 tail | 1 head)
  // a lazy unpack (Group::set_pending_recv): the neighbour rows the last exchange delivered are still in its receive buffer;
  // k_inter mode 0 takes them from there -- nsrc[r]: slot of neighbour row nrows_own + r, -1: not delivered by this exchange --
  // and stores them where an unpack kernel would have (the record array it is given as Znbr)
  const double *recv = nullptr;
  const int *nsrc = nullptr;
};

struct SegTable {
  const Seg *segs = nullptr;
  int nseg_own = 0, nseg_all = 0;
  int rows_own = 0, rows_all = 0;
  const int *own_ptr = nullptr;   // per node: [own_ptr[a], own_ptr[a+1]) own segments
  const int *nbr_ptr = nullptr;   // per node: [nbr_ptr[a], nbr_ptr[a+1]) neighbour segments (indices into segs)
};

// y = A x (+ addv) ; partial[slot] = sum_p < dotv_p , coef * (A x)_p + dotadd_p >
// mode 1 (true): the translation row of x is treated as zero (G_tR R products).  mode 2: y as in mode 1, but the
// dot product sees the full A x -- one pass for "G [0 ; R] + g" and "<x, 1/2 G x + g'>" (DPGOHash.cpp:363-372).
void launch_bsr(int d, hipStream_t st, const SegTable &T, bool all_rows, NodeMask mask, const BsrDev &A,
                const double *x, int mode, const double *addv, double *y, const double *dotv,
                double coef, const double *dotadd, double *partials, int slot,
                // copy1 / copy2: the own rows' records of x are stored there on the way (the tail of iterate(): Xk <- Xak)
                double *copy1 = nullptr, double *copy2 = nullptr);

// y = base + A[:, translation column] t over own rows; tval: the first column of every block of A ((d+1) doubles per
// block), xt: records whose translation row is t.  A quarter of the traffic of launch_bsr.
// mode 1: also out2 = [0 ; Proj_X(y.R)]; mode 2: y not stored, out2 = [0 ; Proj_X(y.R - sym(nabla.R X.R^T) Rdot.R)], the Hessian-vector product (DPGOProblem.cpp:570-574)
// mode 2 with partials: slots 0..3 = <Rdot, out2>, <out2, out2>, <Rdot, Rdot>, <Rdot, rres> (the four scalars of a
// CG step, IterativeSolvers.h:296-347) in the same pass
// mode 1 with partials and dg / dga: slots 0..3 = |out2|^2, <X, y>, <X, dg>, <X, dga> (the start of a refinement: gradient
// norm and f from the model gradient y)
// mode 0 with partials and ds / dgrad / dhs / dg / dga: slots 0..5 = <ds,ds>, <dgrad,ds>, <ds,dhs> (rotation rows),
// <xt,dg>, <xt,dga>, <xt,y> (the sums of a trial point xt, TNT.h:505-536)
// (the epilogue sums always occupy 6 consecutive slots)
void launch_bsr_tcol(int d, hipStream_t st, const SegTable &T, NodeMask mask, const BsrDev &A, const double *tval,
                     const double *xt, const double *base, double *y, int mode = 0, const double *X = nullptr,
                     const double *nabla = nullptr, const double *Rdot = nullptr, double *out2 = nullptr,
                     const double *rres = nullptr, double *partials = nullptr, const double *dg = nullptr,
                     const double *dga = nullptr, const double *ds = nullptr, const double *dgrad = nullptr,
                     const double *dhs = nullptr);

// what the robust inter-edge pass does on the way, instead of a launch of its own
struct InterFuse {
  // mode 0 (update()): Dfobj = G X + g from the product that is there already, its tangent projection and |grad F|^2 into
  // partial slot gn_slot -- k_tangent_full's job (DPGOProblem.cpp:145-162); X: the own rows' records
  const double *GX = nullptr, *X = nullptr;
  double *Df = nullptr;
  int gn_slot = 0;
  // mode 1 (iterate()): the point itself is formed on the way, Z = Zc + gamma (Zc - Zp) -- k_extrapolate's job
  // (DPGOHash.cpp:255-256), for the row and for every pose its incidences reach; the own rows are stored to Yout
  const double *Zc = nullptr, *Zp = nullptr;
  double *Yout = nullptr;
  // mode 1 with the kept products (GXc / GXp: the pass forms Df itself): the proximal half step on the way -- k_proximal's job
  // (DPGOProblem.cpp:600-632): Xout = proximal(Y, Df), |Xout - Xref|^2 into partial slot gn_slot, Xref's rotations <- Xout's;
  // Df itself is then only stored if the caller asks for it
  double *Xout = nullptr, *Xref = nullptr;
  const double *Tinv = nullptr, *Nv = nullptr, *Vb = nullptr;
};
// Robust inter-edge pass (B-form, DPGOProblem.cpp:634-725).
//  mode 0 (update): all rows.  DfE <- B1^T W B1 Z; own rows also g <- DfE - D z.
//     slot 0: sum of rho_e (tail incidences); if quad: slot 1 = sum tr(dZ^T (DfE_old + 1/2 Q dZ)); slot 2: <z, g> over
//     own rows (always written: three consecutive slots).
//  mode 1 (evaluate_g): own rows only, g <- (B1^T W B1 Z)_own - D z.
void launch_inter(int d, hipStream_t st, const SegTable &T, NodeMask mask, const InterEdgesDev &E, int loss,
                  double loss_reg, int mode, bool quad, const double *Z, const double *Zprev,
                  const double *Qdiag, const double *Ddiag, double *DfE, double *g, double *partials,
                  double *wout = nullptr,   // wout (mode 0): the loss weight of every edge
                  // mode 1 with all four: also Df_out = g + GXc + gamma[node] (GXc - GXp) over own rows -- Df at the extrapolated
                  // point from the products G X[k], G X[k-1] the last two update()s left (no pass over G)
                  const double *GXc = nullptr, const double *GXp = nullptr, const NodeCoefs *gamma = nullptr, double *Df_out = nullptr,
                  // mode 0 with Znbr: the neighbour rows are read from Znbr and copied into Z on the way (the halo copy of update())
                  const double *Znbr = nullptr,
                  // gamma_dev: the same gammas in device memory (launch_set_coefs) -- read instead of `gamma` by a launch that may be
                  // replayed from a captured graph, whose by-value arguments are frozen
                  const double *gamma_dev = nullptr,
                  const InterFuse *fuse = nullptr);

// ---- Rescale::Dynamic on the device (see k_rescale_decide / k_rescale_apply) ----
// decide: flags[a] / host_flags[a] = node a (of `nodes`) is rescaled; its scales and counter are updated
void launch_rescale_decide(hipStream_t st, int nnodes, NodeBits nodes, const int *e_off, const double *w, double *scale,
                           int *count, int max_count, int *flags, double *host_flags);
struct RescaleArgs {
  const int *flags = nullptr;
  const double *scale = nullptr, *Gbase = nullptr, *Hbase = nullptr;
  const int *gpos = nullptr;      // 4 ints per own pose: offset of the round's values in Gval, blocks in the round, lane, CSR block index
  const int *att_pos = nullptr;
  double *Gval = nullptr, *Gtcol = nullptr, *Dd = nullptr, *Qd = nullptr, *Tinv = nullptr, *N = nullptr, *V = nullptr, *att_val = nullptr;
  double xi = 0.0;
};
// apply: diagonal blocks of G, D, Q, the proximal coefficients and the diagonal of G_tt of the flagged nodes from the scales
void launch_rescale_apply(int d, hipStream_t st, const SegTable &T, const InterEdgesDev &E, const RescaleArgs &A);

// Objective of every node at Z (own + neighbour rows): partial[slot0] = sum of intra-edge costs,
// partial[slot0 + 1] = sum of rho over inter-edge costs; eform selects the data-matrix form (trivial loss).
void launch_cost(int d, hipStream_t st, const SegTable &T, NodeMask mask, const InterEdgesDev &Ei,
                 const InterEdgesDev &Ee, bool eform, int loss, double loss_reg, const double *Z, double *partials,
                 int slot0);
// partial[slot] = sum |a_p - b_p|^2 over own rows
void launch_sqdist(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *a, const double *b,
                   double *partials, int slot);
// Xout = proximal(Z, Df) per own pose (DPGOProblem.cpp:600-632).  With Xref: partial ||Xout - Xref||^2, after which
// Xref takes over Xout's rotation rows (the next step recovers its translations: DPGOHash.cpp:369-372).
void launch_proximal(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *Z, const double *Df,
                     const double *Tinv, const double *N, const double *V, double *Xout, double *Xref,
                     double *partials, int slot);

// out = a + gamma[node] * (a - b) over all rows (own + neighbour)      (DPGOHash.cpp:255-262)
void launch_extrapolate(int d, hipStream_t st, const SegTable &T, bool all_rows, NodeMask mask,
                        const NodeCoefs &gamma, const double *a, const double *b, double *out, const double *gamma_dev = nullptr);
// the same for three pairs in one launch: (za, zb) -> zout over all rows, (ga, gb) -> gout and (da, db) -> dout over the own rows
void launch_extrapolate3(int d, hipStream_t st, const SegTable &T, NodeMask mask, const NodeCoefs &gamma, const double *gamma_dev,
                         const double *za, const double *zb, double *zout, const double *ga, const double *gb, double *gout,
                         const double *da, const double *db, double *dout);
// dev[a] = C.a[a], a < n: the per-iteration coefficients where replayed launches find them (k_set_coefs)
void launch_set_coefs(hipStream_t st, const NodeCoefs &C, int n, double *dev);
// The tail of iterate() with the exchange's pack on the way: xk = xak (and z = xak, if given) over the own rows of the masked
// nodes, and -- further workgroups of the same launch -- pack[k] = xak[pack_rows[k]], k < npack, whatever the mask (the send
// buffer of the boundary exchange, DPGOHash.h:64-82: own rows of nodes outside the mask have not changed, and xak holds them)
void launch_tail_pack(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *xak, double *xk, double *z,
                      const int *pack_rows, int npack, double *pack);
// out = alpha * a + beta * b  (b may be null); parts: 0 whole record, 1 translation only, 2 rotation only
// out2 (part 0 only): a second copy of the result
void launch_axpby(int d, hipStream_t st, const SegTable &T, bool all_rows, NodeMask mask, double alpha,
                  const double *a, double beta, const double *b, double *out, int part, double *out2 = nullptr);
// out = C.a[node] * a + C.b[node] * b over own rows
void launch_axpby_node(int d, hipStream_t st, const SegTable &T, NodeMask mask, const NodeCoefs &C, const double *a,
                       const double *b, double *out);
// p = -v + cg[node].be * p (the CG direction update with the device-resident beta)
void launch_cg_dir(int d, hipStream_t st, const SegTable &T, NodeMask mask, const CgNode *cg, const double *v, double *p);
// n <= MAX_DOTS dot products in one pass over own rows: partial[slot0 + q] = sum <a_q, b_q> over parts[q]
// (0 whole record, 1 translation row, 2 rotation rows); always writes MAX_DOTS slots
void launch_dots(int d, hipStream_t st, const SegTable &T, NodeMask mask, int n, const double *const *a,
                 const double *const *b, const int *parts, double *partials, int slot0);
// one CG step (IterativeSolvers.h:340-390): s += C.a[node] p, hs += C.a[node] Hp, and r += C.b[node] Hp where C.b != 0
void launch_cg_step(int d, hipStream_t st, const SegTable &T, NodeMask mask, const NodeCoefs &C, const double *p,
                    const double *Hp, double *s, double *hs, double *r, const CgNode *cg = nullptr,   // cg: coefficients from the device state
                    const double *r0 = nullptr,    // r0: first step of a run -- s = hs = 0 (not read), r = r0
                    // xprop: the nodes of *rmask (device) also get xprop.Y = proj_SO(d)(X.Y + s.Y), xprop.x = 0 (launch_retract_rot)
                    const double *X = nullptr, double *xprop = nullptr, const NodeBits *rmask = nullptr);
// start of a truncated CG (IterativeSolvers.h:230-260): s = 0, hs = 0, r = grad, v = pgrad, p = -pgrad; with s == nullptr
// only p = -pgrad (the first launch_cg_step, given r0 = grad, supplies the rest)
void launch_cg_init(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *grad, const double *pgrad,
                    double *s, double *hs, double *r, double *v, double *p);
// gradF = [V.x ; Proj_R(V.Y)] (DPGOProblem.cpp:145-162); partial ||gradF||^2; out may be null
// with add: the vector is V + add, stored to sum_out if given (Dfobj = G X + g from its two halves)
void launch_tangent_full(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *X,
                         const double *V, double *out, double *partials, int slot, const double *add = nullptr,
                         double *sum_out = nullptr);
// dst = src on the neighbour rows only
void launch_copy_nbr_rows(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *src, double *dst);
// out.Y = Proj_R(in.Y), out.x = 0                                       (DPGOProblem.cpp:164-178)
// with dotv: partial[slot] = <dotv.Y, out.Y> in the same pass; two: partial[slot] = |out.Y|^2, partial[slot + 1] = <dotv.Y, out.Y>
// neg: also neg = -out (the first CG direction)
void launch_tangent_rot(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *X,
                        const double *in, double *out, const double *dotv = nullptr, double *partials = nullptr,
                        int slot = 0, bool two = false, double *neg = nullptr);
// out.Y rows = dinv (one entry per rotation row) * in.Y rows: Preconditioner::Jacobi   (DPGOProblem.cpp:96-98, 583-585)
void launch_rot_rowscale(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *dinv, const double *in, double *out);
// out.Y = proj_SO(d)(X.Y + V.Y); out.x = 0                              (SOdProduct.h:111-116)
void launch_retract_rot(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *X,
                        const double *V, double *out);
// dst[didx[k]] = src[sidx[k]] (didx may be null: dst[k]); halo copy, pack, unpack (DPGOHash.h:28-86)
void launch_copy_indexed(int d, hipStream_t st, int count, const int *didx, const int *sidx, const double *src,
                         double *dst, const NodeBits *gate = nullptr);   // gate: a device word; 0 switches the launch off
// partial[slot] = sum_p < x_p , coef * (D_p x_p) + addcoef * add_p > over own rows
void launch_bdiag_dot(int d, hipStream_t st, const SegTable &T, NodeMask mask, const double *Dd, const double *x,
                      double coef, const double *add, double addcoef, double *partials, int slot);

// host_scalars[node * MAX_SLOTS + s] = sum of the node's partials, s < nslots, written straight to pinned host
// memory; *host_flag = seq once all of them are there (arrived: a zeroed device counter)
void launch_reduce(hipStream_t st, const SegTable &T, int nnodes, bool all_rows, int nslots, const double *partials,
                   double *host_scalars, unsigned *arrived, unsigned long long *host_flag, unsigned long long seq,
                   unsigned long long *dev_seq);   // seq == 0: *dev_seq + 1 (a replayed launch); *dev_seq ends up holding the value used

// ---- the gate of a speculative update (k_reduce_gate): the per-node scalars the host knows when it enqueues the gate, by value
struct AmmGate {
  int nnodes = 0, ds = 0, max_it = 0, max_acc = 0, max_hits0 = 0, max_hits1 = 0;
  double sqrt_eps = 0, eta1 = 0, rel_tol = 0, step_tol = 0, psi = 0, phi = 0;
  double f[MAX_LOCAL_NODES], Fk0[MAX_LOCAL_NODES], Fk1[MAX_LOCAL_NODES], fobj[MAX_LOCAL_NODES];
  int hits0[MAX_LOCAL_NODES], hits1[MAX_LOCAL_NODES];
};
// launch_reduce (own rows, nslots sums per node -> host_scalars and dev_scalars) with the gate behind it in the same launch:
// tnt: the refinement's start per node (TNT_SUMMARY each, launch_cg_scal_begin's dev_tnt); *go = ~0 / 0, host_out[0] = 1.0 / 0.0
// (pinned), both set before the flag is raised
void launch_reduce_gate(hipStream_t st, const SegTable &T, int nnodes, int nslots, const double *partials, double *host_scalars,
                        unsigned *arrived, unsigned long long *host_flag, unsigned long long seq, unsigned long long *dev_seq,
                        double *dev_scalars, const AmmGate &G, const double *tnt, const CgNode *cg, NodeBits *go, double *host_out);

// AMM-PGO*'s master sums (k_star_sums): out[0..3] (device) from the partial slots 0..5; launch_publish: n <= 64 device values
// to pinned host memory, then *host_flag = seq
void launch_star_sums(hipStream_t st, const SegTable &T, int nnodes, unsigned valid_slots, const int *slots6, const double *partials,
                      double *out);   // valid_slots: bit q = sum q was produced; slots6[q]: the partial-sum slot it is in
void launch_publish(hipStream_t st, const double *vals, int n, double *host, unsigned long long *host_flag, unsigned long long seq,
                    unsigned long long *dev_seq);

// ---- device-side control of the truncated CG (tnt.cpp) ----
constexpr int CG_SUMMARY = 4;    // doubles per node k_cg_scal writes to pinned memory: stop ordinal, |h|_M, iterations
// Summary word 0 is the ordinal of the scalar step that ended the node's CG (CgNode::stop_ord), or CG_LIVE_ORD while it runs: the
// host asks "was the node live after scalar step w" as word0 > w, and gets the same answer whether it reads the summary of
// step w or that of a later step already written over it -- what it launches next (node sets, tile classes) must not depend
// on how far the stream has run ahead of it.
constexpr double CG_LIVE_ORD = 1e18;
constexpr int TNT_SUMMARY = 8;   // doubles per node k_tnt_begin writes: the six sums it reduced, then `active`
// tnt_begin: the first trust-region iteration's norms, gradient tests and CG start values, all on the device (see k_tnt_begin)
void launch_tnt_begin(hipStream_t st, const SegTable &T, int nnodes, NodeBits bits, bool use_precon, int max_it, double grad_tol,
                      double pgrad_tol, double kappa, double theta, const double *Delta, const double *partials, CgNode *cg,
                      NodeBits *dmask, double *host_tnt);
// tnt_begin and the phase-0 step of the first CG step in one launch (k_cg_scal_begin): the refinement's six sums in the partial
// slots 0..3 and MAX_DOTS.., the step's four from slot cg_first_slot() on; the flag protocol of launch_cg_scal
void launch_cg_scal_begin(hipStream_t st, const SegTable &T, int nnodes, NodeBits bits, bool use_precon, int max_it, double grad_tol,
                           double pgrad_tol, double kappa, double theta, const double *Delta, const double *partials, CgNode *cg,
                           NodeBits *dmask, double *host_tnt, double *host_scalars, unsigned *arrived, unsigned long long *host_flag,
                           unsigned long long seq, unsigned long long *dev_seq, double *dev_tnt = nullptr,   // dev_tnt: host_tnt's numbers in device memory too
                           // carry: the reduction that closes the LAST update() rides along (k_reduce's work: upd_nslots sums per node over
                           // own and neighbour segments of the partial sums from slot UPD_SLOT0 on, to upd_host[node * MAX_SLOTS + s])
                           int upd_nslots = 0, double *upd_host = nullptr);
int cg_first_slot();
// cg_begin: state of the nodes in `bits` from the start values; dmask[0] = dmask[1] = the live ones, dmask[2] = the others.
void launch_cg_begin(hipStream_t st, int nnodes, NodeBits bits, const CgStart &S, int max_it, CgNode *cg, NodeBits *dmask);
// dmask[0]: the nodes of the step under way; dmask[1]: the nodes that go on after it.
// phase 0 (after H p and its four dot products, partial slots 0..3): step length / boundary / negative-curvature
// logic of IterativeSolvers.h:296-362; nodes that stop are cleared from dmask[1].
// phase 1 (after the preconditioner and <r, v>, partial slot 0): beta and the recurrences (:364-390), then the
// stopping test of the next step (:285-291); nodes that stop are cleared from dmask[1], then dmask[0] = dmask[1].
// dmask[2] collects the nodes whose CG has ended (their trial point may be taken).
// Either phase ends by writing, per node, (stop ordinal or CG_LIVE_ORD, h_M_norm, cg_it) to host_scalars[node * CG_SUMMARY + 0..2] and
// raising *host_flag to seq (same protocol as launch_reduce).
// seq == 0 (a launch captured into a graph): the flag is raised to *dev_seq + 1; *dev_seq always ends up holding the value used.
void launch_cg_scal(hipStream_t st, const SegTable &T, int nnodes, int phase, const double *partials, CgNode *cg,
                    NodeBits *dmask, double *host_scalars, unsigned *arrived, unsigned long long *host_flag,
                    unsigned long long seq, unsigned long long *dev_seq);
bool prof_enabled();

// ---- multifrontal SPD solve (spd.h) ----
// One tile of the solve with everything it needs to know about its front: 64 bytes, one load.
struct alignas(16) SpdItem {
  int front, first, count, w;            // first row (forward) / pivot column (backward) of the tile, rows in it
  int u, ld, piv_ptr, upd_ptr;           // ld: doubles between consecutive rows of the tile's panel
  int pos_off, ubuf_off, node, wait_ctr; // node: local node the front belongs to (launch masks)
  int64_t mat_off;                       // offset of the tile's panel
  int wait_need, signal_ctr;             // (reserved: with wait_ctr, the counters of the one-launch solve that round 5 removed)
};
static_assert(sizeof(SpdItem) == 64, "SpdItem is loaded as four int4");

struct SpdDev {
  const int *piv_idx = nullptr, *upd_idx = nullptr;   // matrix indices of the pivots / update rows of every front
  const int *asm_ptr = nullptr;   // per front position p: it receives update-buffer rows asm_ptr[p] .. asm_ptr[p+1]
  const int *ubuf_dst = nullptr;  // per update row of a front (ubuf_off + r): its row in the update buffer
  const double *W = nullptr, *WT = nullptr;           // panels of the backward / forward tiles
  const SpdItem *fwd_items = nullptr, *bwd_items = nullptr;
  const SpdItem *root_items = nullptr;   // the fused root tiles (k_spd_level MODE 2) and their panels
  const double *Wroot = nullptr;
  double *ubuf = nullptr;
};
// One level of the forward / backward sweep.  The tiles of a level are stored node by node -- a node's wide tiles
// (`rows` = 64 or 16 high, one workgroup each, longest first), then its narrow ones (one wave each) -- and a launch
// covers the nodes the host still counts as live: workgroup b works for live slot b % nlive on that node's
// (b / nlive)-th tile, so the grid shrinks with the set of nodes (a masked-out node costs no workgroup at all), a
// node's longest tiles still start first, and with 8 live nodes a node's tiles all run on one XCD (round-robin
// dispatch), next to its vectors in that XCD's L2.  All of it is decided from kernel arguments: no load before the
// tile's own descriptor.
struct SpdLevelMap {
  int nlive = 0, wide_wgs = 0, narrow_wgs = 0, pad = 0;   // wide_wgs = nlive * max wcount, narrow_wgs = nlive * ceil(max ncount / waves); pad: first tile of the level (trace builds)
  int wstart[MAX_LOCAL_NODES], wcount[MAX_LOCAL_NODES];   // per live slot: first wide tile (index into the sweep's items), how many
  int nstart[MAX_LOCAL_NODES], ncount[MAX_LOCAL_NODES];   // the same for the narrow tiles
  unsigned char node[MAX_LOCAL_NODES];                    // local node of the slot
};
// dof = 1: unknown i is the translation of pose i; dof = d: unknown i = (pose i / d, rotation row i % d).
// vec is a record array: forward reads the right-hand side from vec and writes y to ytmp (n x d, the pivots
// of a front consecutive); backward reads ytmp and writes scale * A^-1 b into vec (scale must be +1 or -1).
// Panels of the solve tiles from a factor that is still on the device (front-major W / WT, spd_dev.hip): tile i copies
// len rows of `count` entries, src_ld apart, starting at src + src_off, to panels + items[i].mat_off (rows items[i].ld apart).
struct PanelSrc { long long src_off; int src_ld; int len; };
void launch_pack_panels(hipStream_t st, const SpdItem *items, const PanelSrc *srcs, int ntiles, const double *src, double *panels);
// measurement builds (-DSPD_TRACE) only: where the solve tiles write their phase timestamps (6 per tile); no-op otherwise
void spd_trace_set(unsigned long long *p);
int spd_waves(int rows);   // waves (= narrow tiles) per workgroup of the class with `rows`-high wide tiles
// mode 0: a forward level; 1: a backward level; 2: the roots of the trees, forward and backward step in one pass (vec: the
// right-hand side records, ytmp: the records that receive scale * A^-1 b on the roots' unknowns; must not be the same array)
void launch_spd_level(int d, int dof, hipStream_t st, const SpdDev &S, int mode, const SpdLevelMap &M, int rows,
                      double *vec, double *ytmp, double scale, double level_bytes = 0.0, bool stream_once = true,
                      NodeMask mask = ALL_NODES);   // mask.p: the device-side mask (nodes that stopped since the host last looked)
// the dense w x w products L11^-T L11^-1 of the root fronts (k_root_syrk): front i reads its W_s (w x ld) at src + src_off and
// writes at dst + dst_off; the fused root tiles' panels are then cut out of dst with launch_pack_panels
struct RootDesc { long long src_off, dst_off; int ld, w; };
void launch_root_syrk(hipStream_t st, const RootDesc *rd, int nroots, int max_w, const double *src, double *dst);

// ---- the roots as ONE TRIANGLE (k_root_sym + k_root_combine) ----
// The product P = L11^-T L11^-1 a fused root streams is symmetric: storing its lower triangle in 64 x 64 blocks halves the
// bytes of the level.  Block (I, J), J <= I, is stored k-major ([k][r]: entry (I*64 + r, J*64 + k); 4096 doubles, rows and
// columns past w are zero) and serves BOTH x_I += P_IJ f_J (lane = row, plain FMAs) and, for J < I, x_J += P_IJ^T f_I (the
// block's 8-column strips go through LDS so that lane = column can take its dot product down the rows).  The diagonal
// blocks are stored in full and take the first product only.  An ITEM (one workgroup of four waves; SpdItem with first = I*64, count = rows
// of block row I, u = J0, ld = number of blocks, upd_ptr = its direct slot, ubuf_off = its first transposed slot,
// mat_off = its blocks, consecutive) covers blocks (I, J0 .. J0 + ld - 1) and writes 64 x D partial sums per output block:
// one "direct" slot and one "transposed" slot per off-diagonal block.  k_root_combine then adds the slots of every block
// row in a FIXED order -- no atomics, results independent of the schedule -- scales and writes the solution records.
struct alignas(16) RootRow {
  int piv_ptr, first, count, node;   // block row R of a root front: pivots piv_ptr + first .. + count, local node
  int dslot, ndslots, tbase, nb;     // its direct slots; transposed slot of block (I, R), I > R: tbase + I (I - 1) / 2 + R
  int R, pad0, pad1, pad2;
};
static_assert(sizeof(RootRow) == 48, "RootRow is loaded as three int4");
constexpr int ROOT_SYM_MAXJ = 8;     // blocks per item at most
// items: M.nstart / ncount (one workgroup each) index S.root_items; part: the partial-sum slots
void launch_root_sym(int d, int dof, hipStream_t st, const SpdDev &S, const SpdLevelMap &M, const double *vec, double *part,
                     double bytes, bool stream_once, NodeMask mask);
// rows: M.wstart / wcount index `rows`; out receives scale * (sum of the slots) on the roots' unknowns
void launch_root_combine(int d, int dof, hipStream_t st, const SpdDev &S, const SpdLevelMap &M, const RootRow *rows,
                         const double *part, double scale, double *out, NodeMask mask);

// ---- optional per-launch timing (HIP events on the launch stream), off by default ----
enum ProfKind { PK_BSR = 0, PK_INTER, PK_PROX, PK_AXPBY, PK_DOT, PK_ROTOP, PK_COPYIDX, PK_BDIAG, PK_REDUCE,
                PK_SPD_FWD, PK_SPD_BWD, PK_BSR_TCOL, PK_COUNT };
void prof_enable(bool on);
// One timing scope around the back-to-back launches of a solve sweep (profiling pass only; nothing otherwise)
struct ProfSweep {
  struct Impl;
  Impl *p;
  ProfSweep(bool forward, hipStream_t st, double bytes, int launches);
  ~ProfSweep();
  ProfSweep(const ProfSweep &) = delete;
  ProfSweep &operator=(const ProfSweep &) = delete;
};
void prof_reset();
void prof_collect(double *ms, double *bytes, long *count);
// ... and, for the fused passes (k_inter, k_proximal), the bytes of every operand the pass has to move, counted one by one
// (DESIGN 7a): SURVEY 8(d)'s formula prices a bare residual pass, the kernels also carry the surrogate's per-pose blocks
void prof_collect_operands(double *operand_bytes);

}  // namespace dpgo
