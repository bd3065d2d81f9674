// Riemannian truncated-Newton trust-region refinement on device vectors,
// batched over the nodes of a group.
//
// Same algorithm as the reference's generic solvers
//   TNT    C++/Optimization/include/Optimization/Riemannian/TNT.h:242-693
//   STPCG  C++/Optimization/include/Optimization/LinearAlgebra/IterativeSolvers.h:166-426
// instantiated the way DPGOHash does (C++/DPGO/src/DPGOHash.cpp:270-349):
//   f(Y)        = G(Y | g, f)                       DPGOProblem.cpp:180-205
//   grad        = Proj_R(g_R + (G Y)_R)             DPGOProblem.h:380-406
//   Hess[Rdot]  = Proj_R(G_Rt tdot + G_RR Rdot - SBD(Rdot, R, nabla)), tdot = -G_tt^-1 G_tR Rdot
//                                                   DPGOProblem.cpp:552-577
//   precon      = Proj_R((G_RR + lambda I)^-1 v)    DPGOProblem.cpp:579-598
//   retraction  = [ -G_tt^-1(g_t + G_tR R+) ; R+ = proj(R + V) ]   DPGOProblem.cpp:127-143
//   metric      = sum V1 .* V2                      DPGOHash.cpp:307-310
//
// Every node runs the reference's control flow with its OWN scalars (step
// lengths, trust-region radius, stopping tests); the nodes advance in lockstep so
// that one launch serves all of them.  Per-node step lengths and the set of nodes
// a launch works on travel as kernel arguments (NodeCoefs, NodeMask: no uploads),
// and the handful of reductions per CG step come back through one polled read-back.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "group.h"

namespace dpgo {

#define HIP_CHECK(x)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      fprintf(stderr, "[dpgo_amd] ERROR: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      throw DeviceError(hipGetErrorString(e_));                                                   \
    }                                                                                             \
  } while (0)

namespace {
enum { ST_GRADIENT = 0, ST_PRECON_GRADIENT, ST_REL_DECREASE, ST_STEPSIZE, ST_TRUST_REGION, ST_ITER_LIMIT };

struct NodeTnt {
  bool active = true;       // outer trust-region loop still running
  int status = ST_ITER_LIMIT;
  double fx = 0, gnorm = 0, pgnorm = 0, Delta = 1.0;   // TNTParams::Delta0 (TNT.h:81)
  int iteration = 0, accepted = 0, inner_total = 0;
  // STPCG state (IterativeSolvers.h:207-283)
  bool cg = false;
  double sk_M_pk = 0, sk_M_2 = 0, pk_M_2 = 0, rv = 0, Delta_2 = 0, target = 0, h_M_norm = 0;
  int cg_it = 0;
};
}  // namespace

// The CG steps of a group go out as one graph replay each (a step is 13-21 launches with fixed arguments): wherever the
// segments of the iteration are replayed (Group::segment, iter_graph_wanted), and -- round 4's default, measured +8..13 % on
// city10000 -- from the start for groups of at least two nodes and at most 40 000 poses, whose steps are bound by the host's
// launch rate.  DPGO_CG_GRAPH=0 keeps just these eager (A/B hook), DPGO_ITER_GRAPH=0 everything.
bool Group::cg_graph_wanted() const {
  static const int force = [] { const char *e = getenv("DPGO_CG_GRAPH"); return e ? atoi(e) : -1; }();
  static const int iter_force = [] { const char *e = getenv("DPGO_ITER_GRAPH"); return e ? atoi(e) : -1; }();
  if (force == 0 || iter_force == 0 || graphs_broken_ || prof_enabled()) return false;
  if (iter_graph_wanted()) return true;
  return P0_ <= 40000 && num_local() >= 2;
}

bool Group::run_tnt(const std::vector<int> &nodes, double *X, const double *g, const double *g_alt, bool base_ready,
                    const std::function<bool()> *confirm) {
  if (!confirm) finish_update();
  const Options &o = opt_;
  const int L = num_local();
  const bool jacobi = (o.preconditioner == 1) && jacobi_.n > 0;       // Preconditioner::Jacobi
  const bool use_precon = jacobi || ((o.preconditioner == 3) && Lrr_.F.n > 0);   // ... or RegularizedCholesky
  double *nabla = tmp_[0].p, *grad = tmp_[1].p, *sk = tmp_[2].p, *rk = tmp_[3].p, *vk = tmp_[4].p, *pk = tmp_[5].p,
         *Hp = tmp_[6].p, *xprop = tmp_[7].p, *w1 = tmp_[8].p, *pg = tmp_[10].p, *hh = tmp_[11].p,
         *w3 = tmp_[12].p, *nprop = tmp_[13].p;
  // hh accumulates H s_k alongside s_k (every step s_k += c p_k is mirrored by hh += c H p_k), so the
  // predicted decrease needs no extra Hessian-vector product: same value as Hess(x, h) of TNT.h:514-515
  // up to rounding of the CG recurrence.
  std::vector<NodeTnt> S(L);
  for (auto &s : S) s.active = false;
  for (int a : nodes) S[a] = NodeTnt();

  const int P2[MAX_DOTS] = {2, 2, 2, 2, 2, 2};
  // nabla = G Y + g and grad = Proj_Y(nabla) (rotation rows).  from_base: Y.t was just recovered from Y.R with
  // this g (recover_translations), so T1_ = G [0 ; Y.R] + g is there and only the translation column is missing.
  // Returns true when the pass also left the four sums of the refinement's start (|grad|^2, <Y, nabla>, <Y, g>,
  // <Y, g_alt>) in the partial slots 0..3 (its epilogue), so that no separate pass over the vectors is needed.
  const double *ga = g_alt ? g_alt : g;
  auto quad_model = [&](const double *Y, bool from_base) {
    if (from_base) {
      apply_tcol(Y, T1_.p, nabla, 1, Y, nullptr, nullptr, grad, nullptr, partials_.p, g, ga);   // nabla, grad and the sums in one pass
      return true;
    }
    launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, Y, false, g, nabla, nullptr, 0, nullptr, nullptr, 0);
    launch_tangent_rot(d_, st_, T_, cur_mask_, Y, nabla, grad);
    return false;
  };
  // out = P(v) = Proj_Y(M^-1 v) with |out|^2 and <v, out> in the partial slots MAX_DOTS, MAX_DOTS + 1, and -out in pk
  // (the first CG direction); only called with a preconditioner
  auto precon_with_sums = [&](const double *Y, const double *v, double *out) {
    if (jacobi) launch_rot_rowscale(d_, st_, T_, cur_mask_, jacobi_.p, v, w1);
    else solve_rr(const_cast<double *>(v), w1, 1.0);   // w1.R = (G_RR + lambda I)^-1 v.R; the forward sweep only reads v
    launch_tangent_rot(d_, st_, T_, cur_mask_, Y, w1, out, v, partials_.p, MAX_DOTS, true, pk);
  };
  // gnorm, pgnorm (and rv0 = <grad, P grad>, the first CG scalar) of the nodes in `set` (mask == set).
  // with_f: f(X | g) in the same read-back, from the model gradient nabla = G X + g that is there anyway:
  //   f = <X, g> + 1/2 <X, G X> = 1/2 (<X, nabla> + <X, g>)      (DPGOProblem.cpp:180-205)
  // lin / lin_alt = <X, g> / <X, g_alt> let the caller re-base f on g_alt without another G X.
  std::vector<double> rv0(L, 0.0), lin(L, 0.0), lin_alt(L, 0.0);
  // the dot products (partial slots 0..3 and MAX_DOTS, MAX_DOTS + 1) and the first CG direction pk = -P(grad); who
  // reduces the sums is the caller's choice: k_reduce + wait (norms) or k_tnt_begin (no wait).
  // have_sums: slots 0..3 were already left there by quad_model's epilogue
  auto norms_enqueue = [&](bool with_f, bool have_sums) {
    if (!have_sums) {
      const double *pa[MAX_DOTS] = {grad, X, X, X}, *pb[MAX_DOTS] = {grad, nabla, g, ga};
      const int parts[MAX_DOTS] = {2, 0, 0, 0, 0, 0};
      launch_dots(d_, st_, T_, cur_mask_, with_f ? 4 : 1, pa, pb, parts, partials_.p, 0);
    }
    if (use_precon) precon_with_sums(X, grad, pg);
    else launch_cg_init(d_, st_, T_, cur_mask_, grad, grad, nullptr, nullptr, nullptr, nullptr, pk);
  };
  auto norms_take = [&](int a, bool with_f, double g2, double xn, double xg, double xga, double pg2, double gpg) {
    S[a].gnorm = S[a].pgnorm = std::sqrt(g2);
    rv0[a] = g2;
    if (use_precon) {
      S[a].pgnorm = std::sqrt(pg2);
      rv0[a] = gpg;
    }
    if (with_f) {
      S[a].fx = 0.5 * (xn + xg) + res_[a].f;
      lin[a] = xg;
      lin_alt[a] = xga;
    }
  };
  // ... from the pinned summary of k_tnt_begin
  auto norms_read = [&](const std::vector<int> &set, bool with_f) {
    for (int a : set) {
      const double *t = h_tnt_ + a * TNT_SUMMARY;
      norms_take(a, with_f, t[0], t[1], t[2], t[3], t[4], t[5]);
    }
  };
  auto norms = [&](const std::vector<int> &set, bool with_f, bool have_sums) {
    norms_enqueue(with_f, have_sums);
    fetch(MAX_DOTS + 2, false);
    for (int a : set) norms_take(a, with_f, scal(a, 0), scal(a, 1), scal(a, 2), scal(a, 3), scal(a, MAX_DOTS), scal(a, MAX_DOTS + 1));
  };

  const double sqrt_eps = std::sqrt(std::numeric_limits<double>::epsilon());
  const double eta1 = .05, eta2 = .9, alpha1 = .25, alpha2 = 2.5, Delta_tol = 1e-6;   // TNT.h:83-97,129
  constexpr int NSUM = 6;   // the sums a trial point needs (TNT.h:505-536)
  // trial point of the nodes in `m`: x+ = retract(x, s), f(x+) and, for an accepted step, the next model gradient
  // retracted: the rotations of x+ are there already (the CG step's vector update took them along, stepA)
  auto enqueue_trial = [&](NodeMask m, bool retracted, int nslots, bool with_reduce = true) {
    cur_mask_ = m;
    if (!retracted) launch_retract_rot(d_, st_, T_, cur_mask_, X, sk, xprop);
    recover_translations(xprop, g);
    // nprop = G xprop + g: gives f(xprop) and, if accepted, the next model; its epilogue leaves the six sums
    // <s,s>, <grad,s>, <s,Hs>, <x+,g>, <x+,g_alt>, <x+,nprop> in the partial slots 0..5
    apply_tcol(xprop, T1_.p, nprop, 0, nullptr, nullptr, nullptr, nullptr, nullptr, partials_.p, g, ga, sk, grad, hh);
    // (with_reduce = false: the caller launches the reduction itself, with the gate of a speculative update: group.h)
    if (with_reduce) launch_reduce(st_, T_, L, false, nslots, partials_.p, h_scal_, reduce_arrived_.p, h_flag_, next_seq(), dev_seq_.p);
  };
  // acceptance test and trust-region update of node a from the sums of its trial point (TNT.h:537-607)
  std::vector<int> acc, requad;
  auto judge = [&](int a, const double *t) {
    NodeTnt &s = S[a];
    const double fx_prop = 0.5 * (t[5] + t[3]) + res_[a].f;
    const double h_norm = std::sqrt(t[0]);
    const double dm = -t[1] - 0.5 * t[2];
    const double df = s.fx - fx_prop;
    const double rel_dec = df / (sqrt_eps + std::fabs(s.fx));
    const double rho = df / dm;
    const bool ok = (!std::isnan(rho)) && rho > eta1;
    s.accepted += ok;
    bool stop = false;
    if (ok) {
      acc.push_back(a);
      s.fx = fx_prop;
      lin[a] = t[3];
      lin_alt[a] = t[4];
      if (rel_dec < o.rel_func_decrease_tol) { s.status = ST_REL_DECREASE; stop = true; }
      else if (h_norm < o.stepsize_tol) { s.status = ST_STEPSIZE; stop = true; }
      else if (s.iteration + 1 < o.max_iterations && s.accepted < o.max_iterations_accepted)
        requad.push_back(a);   // the new model is only needed if another iteration follows (TNT.h:446-449)
    }
    if (!stop) {   // trust-region update (TNT.h:593-607)
      if ((!std::isnan(rho)) && rho >= eta2) s.Delta = std::max(alpha2 * s.h_M_norm, s.Delta);
      else if (std::isnan(rho) || rho < eta1) {
        s.Delta = alpha1 * s.h_M_norm;
        if (s.Delta < Delta_tol) { s.status = ST_TRUST_REGION; stop = true; }
      }
    }
    if (stop) s.active = false;
    else s.iteration++;
  };
  auto cgs = [&](int a, int k) { return h_cg_[a * CG_SUMMARY + k]; };
  // ---- STPCG (IterativeSolvers.h:207-426).  The scalar recurrences (alpha, beta, the boundary / negative
  // curvature / kernel tests, the stopping test) run on the device (k_cg_scal); the vector kernels take their
  // step lengths and the set of still-iterating nodes from device memory, so a whole CG step is enqueued
  // without a host round trip.  The host only polls the summary (live, |h|_M, iterations) of a step it enqueued
  // earlier: step i+1 is already queued when the outcome of step i arrives; once every node has stopped, the
  // kernels of the surplus step find an empty device mask and return at once.
  // by value: the nodes the host last saw iterating (it sizes the launches -- the solves shrink their grids with it);
  // by pointer: the device's own, more recent masks
  NodeBits bitsA = 0;
  NodeMask mA = ALL_NODES, mB = ALL_NODES;
  // first half of a step: H p and its four scalars, then the step-length logic (:296-362)
  // first: the first step of a run -- s_0 = 0, H s_0 = 0, r_0 = grad are not materialised, the step takes them as given, and
  // it runs for every node of the run, live or not: a node that stops before its first step has c1 = 0 and gets its
  // s = H s = 0 written here
  // retract: the nodes whose CG ends with this step (dmask[2]) get the rotations of their trial point from the vector
  // update (k_cg_step) instead of a launch of their own.  begin: the start of the refinement -- the norms, the gradient tests,
  // the CG's start values (k_tnt_begin) -- has not been taken yet and rides with this step's scalar kernel (k_cg_scal_begin):
  // the product then runs for every candidate, and leaves its sums where the refinement's are not
  auto stepA = [&](bool first, bool retract = false, const std::function<void(const double *)> *begin = nullptr) {
    cur_mask_ = begin ? live_mask(bitsA, nullptr) : mA;
    launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, pk, true, nullptr, w1, nullptr, 0, nullptr, nullptr, 0);   // G [0 ; p.R]
    solve_tt(w1, w3, -1.0);
    double *sums = partials_.p + (begin ? (size_t)cg_first_slot() * T_.nseg_all : 0);
    apply_tcol(w3, w1, nullptr, 2, X, nabla, pk, Hp, first ? grad : rk, sums);   // Hp and <p,Hp>, <Hp,Hp>, <p,p>, <p,r>
    // the step-length logic
    if (begin) (*begin)(sums);
    else launch_cg_scal(st_, T_, L, 0, partials_.p, cg_.p, dmask_.p, h_cg_, reduce_arrived_.p, h_flag_, next_seq(), dev_seq_.p);
    // s += c1 p, H s += c1 H p for every node of the step (a node that stops here takes its boundary step), r += alpha H p
    // for those that go on
    launch_cg_step(d_, st_, T_, first ? NodeMask{bitsA, nullptr} : mA, NodeCoefs(), pk, Hp, sk, hh, rk, cg_.p, first ? grad : nullptr,
                   retract ? X : nullptr, retract ? xprop : nullptr, retract ? dmask_.p + 2 : nullptr);
  };
  // second half: preconditioner; beta and the recurrences, next stopping test (:364-390, :285-291)
  auto stepB = [&]() {
    cur_mask_ = mB;
    if (use_precon) {
      if (jacobi) launch_rot_rowscale(d_, st_, T_, cur_mask_, jacobi_.p, rk, w1);
      else solve_rr(rk, w1, 1.0);
      launch_tangent_rot(d_, st_, T_, cur_mask_, X, w1, vk, rk, partials_.p, 0);   // v = Proj(M^-1 r) and <r, v>
    } else {
      copy_rows(vk, rk, false, 0);
      const double *pa[MAX_DOTS] = {rk}, *pb[MAX_DOTS] = {vk};
      launch_dots(d_, st_, T_, cur_mask_, 1, pa, pb, P2, partials_.p, 0);
    }
    launch_cg_scal(st_, T_, L, 1, partials_.p, cg_.p, dmask_.p, h_cg_, reduce_arrived_.p, h_flag_, next_seq(), dev_seq_.p);
    launch_cg_dir(d_, st_, T_, cur_mask_, cg_.p, vk, pk);
  };
  // what a segment's key must hold beside the rotating buffers (segment()): the vectors of this call and its variant
  // (X is looked up at the time of use: an accepted step swaps the iterate's buffers)
  auto K = [](const void *q) { return (unsigned long long)(uintptr_t)q; };
  const unsigned long long kg = K(g), kga = K(ga), kvar = (use_precon ? 1ull : 0ull) | (jacobi ? 2ull : 0ull) | (base_ready ? 4ull : 0ull);
  // One whole step (A then B, not the first) as ONE submission: captured once per set of argument values, replayed ever
  // after.  The by-value node sets of a replay are the group's nodes -- the device's own masks keep the nodes that are
  // not (or no longer) part of the CG out, as they do for a node that stopped since the host last looked.
  auto graph_step = [&]() {
    const NodeBits all = L >= 64 ? ~0ull : ((1ull << L) - 1);
    const NodeMask sA = mA, sB = mB;
    // (the roots' tile classes are the eager step's: they are part of the arithmetic, and so of the key)
    const unsigned long long cls = (Ltt_.fine_root_for(sA.v) ? 1ull : 0ull) | (use_precon && !jacobi && Lrr_.fine_root_for(sB.v) ? 2ull : 0ull);
    segment(21, all, {K(X), kvar, cls}, [&] {
      mA = NodeMask{all, dmask_.p};
      mB = NodeMask{all, dmask_.p + 1};
      struct Classes {
        Group *g;
        ~Classes() { g->class_tt_ = g->class_rr_ = nullptr; }
      } classes{this};
      class_tt_ = &sA.v;
      class_rr_ = &sB.v;
      stepA(false);
      stepB();
    }, 1);
    mA = sA; mB = sB;
  };
  const bool use_graph = cg_graph_wanted();
  // The nodes still iterating after scalar step `w` of this run (phase 0 / 1 of step j: 2 j - 1 / 2 j).  The summary of a
  // later step may already have overwritten the one waited for; it carries the ordinal each node stopped at (kernels.h,
  // CG_LIVE_ORD), so the answer -- and with it the node sets and tile classes of the next launches -- is the same however
  // late the host comes.
  std::vector<int> A;
  auto live_after = [&](int a, int w) { return cgs(a, 0) > (double)w; };
  auto any_live = [&](int w) {
    NodeBits live = 0;
    for (int a : A)
      if (live_after(a, w)) live |= 1ull << a;
    mA = live_mask(live, dmask_.p);       // (few live nodes: the own-segment launches cover them alone)
    mB = live_mask(live, dmask_.p + 1);
    return live != 0;
  };

  // The first trust-region iteration starts without a host round trip: the norms, the gradient tests and the start
  // values of the CG are taken on the device (k_tnt_begin); the host reads the same sums at its first wait below.
  const bool device_start = o.max_iterations > 0 && o.max_iterations_accepted > 0;
  // In the early regime every node ends its first CG step on the trust-region boundary, so -- as long as that was the case
  // the last time -- the trial point of the nodes whose CG is over (dmask[2]) is enqueued right behind that step and ONE
  // wait brings the norms, the CG summary and the trial point's sums.  Otherwise the step is awaited at once.
  const bool spec = device_start && tnt_speculate_;
  set_mask(nodes);
  const NodeBits bits_nodes = cur_mask_.v;
  unsigned long long seqA = 0, seq_first = 0;
  bool merged = false;
  tnt_common_ = false;
  int tr_rounds = 0;
  if (device_start) {
    // Everything from the model gradient to the first wait is branch-free: ONE segment (a replay where the host's launch
    // rate would bound it).  Every node of `nodes` is a candidate (the iteration limits allow a first iteration), the radii
    // are TNTParams::Delta0.
    bitsA = bits_nodes;
    int nslots = 0;
    if (spec) {
      nslots = std::max((int)NSUM, deferred_slots_);
      deferred_slots_ = 0;
    }
    // where every node of the group is in here and the trial point follows unasked, the update() that the common outcome
    // leads to goes out as well, under a gate that takes the decision on the device (group.h: SpecUpdate); the trial point's
    // reduction then waits for the gate's inputs (below)
    const bool plan_spec = confirm && spec && fused_ && (int)nodes.size() == L && base_ready && X == Xak_.p && spec_update_possible(xprop);
    segment(20, bits_nodes, {K(X), kg, kga, kvar, spec ? 1ull : 0ull, (unsigned long long)nslots, plan_spec ? 1ull : 0ull}, [&] {
      cur_mask_ = live_mask(bits_nodes, nullptr);
      const bool have_sums = quad_model(X, base_ready);
      norms_enqueue(true, have_sums);
      std::vector<double> Delta(L, 0.0);
      for (int a : nodes) Delta[a] = S[a].Delta;
      const std::function<void(const double *)> begin = [&](const double *) {
        // (update()'s reduction, if it is still waiting for somebody to take it along: group.h, UpdLazy)
        const int carry = (upd_lazy_.pending && !capturing_) ? upd_lazy_.nslots : 0;
        launch_cg_scal_begin(st_, T_, L, bitsA, use_precon, o.max_tCG_iterations, o.grad_norm_tol, o.preconditioned_grad_norm_tol,
                             o.STPCG_kappa, o.STPCG_theta, Delta.data(), partials_.p, cg_.p, dmask_.p, h_tnt_, h_cg_, reduce_arrived_.p,
                             h_flag_, next_seq(), dev_seq_.p, dev_tnt_.p, carry, h_upd_);
        if (carry) { upd_lazy_.pending = false; pending_seq_ = fetch_seq_; }
      };
      merged = fused_;
      if (!merged)
        launch_tnt_begin(st_, T_, L, bitsA, use_precon, o.max_tCG_iterations, o.grad_norm_tol, o.preconditioned_grad_norm_tol,
                         o.STPCG_kappa, o.STPCG_theta, Delta.data(), partials_.p, cg_.p, dmask_.p, h_tnt_);
      mA = live_mask(bitsA, dmask_.p);
      mB = live_mask(bitsA, dmask_.p + 1);
      stepA(true, spec && fused_, merged ? &begin : nullptr);
      if (spec) enqueue_trial(NodeMask{bitsA, dmask_.p + 2}, fused_, nslots, !plan_spec);
    });
    mA = live_mask(bitsA, dmask_.p);   // (a replay does not run the body: the host's copies)
    mB = live_mask(bitsA, dmask_.p + 1);
    cur_mask_ = live_mask(bits_nodes, nullptr);
    seq_first = fetch_seq_;   // (the flag of the last launch of the segment: the trial point's reduction, or the first step's scalars)
    // the caller's read-back (the scalars that decide whether these nodes are refined at all) is taken NOW, with the start
    // of the refinement already on the GPU: the stream never waits for that decision
    if (confirm && !(*confirm)()) return false;
    if (plan_spec) {
      seqA = seq_first;                      // (the first step's scalars: the segment's last flag)
      speculate_update(xprop, nslots);       // the trial point's reduction with the gate, then the continuation
      seq_first = spec_upd_.seq_trial;
    } else seqA = seq_first - (spec ? 1 : 0);
  } else {
    flush_deferred();   // (launches that were waiting for this refinement's first segment: there is none on this path)
    if (confirm && !(*confirm)()) return false;
    const bool have_sums = quad_model(X, base_ready);
    norms(nodes, true, have_sums);
  }

  for (bool first_iteration = true;; first_iteration = false) {
    const bool dev = first_iteration && device_start;
    // ---- nodes that start another trust-region iteration (TNT.h:446-484); with `dev` the gradient tests are the
    // device's, and A holds the candidates until the host has seen the summary
    A.clear();
    for (int a : nodes) {
      NodeTnt &s = S[a];
      if (!s.active) continue;
      if (!(s.iteration < o.max_iterations && s.accepted < o.max_iterations_accepted)) { s.active = false; continue; }
      if (!dev) {
        if (s.gnorm < o.grad_norm_tol) { s.status = ST_GRADIENT; s.active = false; continue; }
        if (s.pgnorm < o.preconditioned_grad_norm_tol) { s.status = ST_PRECON_GRADIENT; s.active = false; continue; }
      }
      A.push_back(a);
    }
    if (A.empty()) break;
    tr_rounds++;
    if (tr_rounds > 1) tnt_common_ = false;   // (a further round: not the common course)
    std::vector<double> tsum((size_t)L * NSUM, 0.0);
    std::vector<char> tried(L, 0);
    if (!dev) {
      set_mask(A);
      bitsA = cur_mask_.v;
      // p_0 = -v_0, v_0 = P(grad): written by the pass that took the preconditioned gradient norm (norms_enqueue).
      // A later iteration of a node whose step was rejected starts from the same gradient: p_0 again (pk was overwritten)
      if (!first_iteration) launch_cg_init(d_, st_, T_, cur_mask_, grad, use_precon ? pg : grad, nullptr, nullptr, nullptr, nullptr, pk);
      CgStart cs;
      for (int a = 0; a < L; a++) cs.rv[a] = cs.Delta[a] = cs.target[a] = 0.0;
      for (int a : A) {
        cs.rv[a] = rv0[a];   // <r_0, v_0> = <grad, P grad>, read back together with the norms
        cs.Delta[a] = S[a].Delta;
        const double r0 = std::sqrt(rv0[a]);
        cs.target[a] = r0 * std::min(o.STPCG_kappa, std::pow(r0, o.STPCG_theta));
      }
      launch_cg_begin(st_, L, bitsA, cs, o.max_tCG_iterations, cg_.p, dmask_.p);
      mA = live_mask(bitsA, dmask_.p);
      mB = live_mask(bitsA, dmask_.p + 1);
      stepA(true);
      seqA = fetch_seq_;
    }
    // The first step (enqueued above, or -- with `dev` -- in front of the loop).
    if (dev && spec) wait_flag(seq_first);   // (the trial point's reduction: everything before it is there too)
    else wait_flag(seqA);
    if (dev) {
      // the sums k_tnt_begin reduced, and its verdict on the gradient tests
      norms_read(A, true);
      std::vector<int> act;
      for (int a : A) {
        if (h_tnt_[a * TNT_SUMMARY + 6] != 0.0) { act.push_back(a); continue; }
        S[a].status = S[a].gnorm < o.grad_norm_tol ? ST_GRADIENT : ST_PRECON_GRADIENT;
        S[a].active = false;
      }
      A.swap(act);
    }
    if (dev && spec)
      for (int a : A)
        if (!live_after(a, 1)) {   // its CG ended with (or before) the first step: the sums just read are its trial point's
          tried[a] = 1;
          for (int q = 0; q < NSUM; q++) tsum[(size_t)a * NSUM + q] = scal(a, q);
          S[a].h_M_norm = cgs(a, 1);
          S[a].cg_it = (int)cgs(a, 2);
        }
    const bool more_steps = any_live(1);
    if (dev) tnt_speculate_ = !more_steps;   // speculate next time if nobody needed a second step this time
    if (more_steps) {
      stepB();
      unsigned long long seqB = fetch_seq_;
      for (int w = 2;; w += 2) {
        if (use_graph) graph_step();
        else { stepA(false); stepB(); }
        const unsigned long long next = fetch_seq_;
        wait_flag(seqB);   // the outcome of the step before the one just enqueued (its phase 1: scalar step w)
        if (!any_live(w)) break;
        seqB = next;
      }
    }
    std::vector<int> rest;
    for (int a : A) {
      if (tried[a]) continue;
      rest.push_back(a);
      S[a].h_M_norm = cgs(a, 1);
      S[a].cg_it = (int)cgs(a, 2);
    }
    for (int a : A) {
      S[a].cg = false;
      S[a].inner_total += S[a].cg_it;
    }
    // ---- trial point (TNT.h:505-536) of the nodes that have not had theirs
    if (!rest.empty()) {
      set_mask(rest);
      const NodeMask mrest = cur_mask_;
      int nslots = std::max((int)NSUM, deferred_slots_);
      deferred_slots_ = 0;
      segment(22, mrest.v, {K(X), kg, kga, (unsigned long long)nslots}, [&] { enqueue_trial(mrest, false, nslots); });
      wait_flag(fetch_seq_);
      for (int a : rest)
        for (int q = 0; q < NSUM; q++) tsum[(size_t)a * NSUM + q] = scal(a, q);
    }
    acc.clear();
    requad.clear();
    for (int a : A) judge(a, &tsum[(size_t)a * NSUM]);
    if ((int)acc.size() == L && X == Xak_.p && xprop == tmp_[7].p) {
      // every node of the group took its step: the trial buffer simply becomes the iterate (no copy)
      Xak_.swap(tmp_[7]);
      X = Xak_.p;
      xprop = tmp_[7].p;
      // the common course (group.h: SpecUpdate): the first round, every node's trial point taken behind its first CG step,
      // every step accepted, no further round
      bool all_tried = dev && spec;
      for (int a = 0; a < L; a++) all_tried = all_tried && tried[a];
      tnt_common_ = all_tried && tr_rounds == 1 && requad.empty();
    } else if (!acc.empty()) {
      set_mask(acc);
      copy_rows(X, xprop, false, 0);
    }
    if (!requad.empty()) {
      set_mask(requad);
      copy_rows(nabla, nprop, false, 0);   // the model gradient at the accepted point
      launch_tangent_rot(d_, st_, T_, cur_mask_, X, nabla, grad);
      norms(requad, false, false);
    }
  }
  // scalars the caller parked in the partial sums (amm: the half step's three) ride with the first read-back of this
  // function; if there was none (every node left at the gradient tests), fetch them now
  if (deferred_slots_) fetch(deferred_slots_, false);
  if (o.verbose) {
    static const char *names[] = {"gradient", "preconditioned gradient", "relative decrease", "step size", "trust region", "iteration limit"};
    for (int a : nodes)
      printf("[dpgo_amd] node %d TNT: f = %.12e, |grad| = %.3e, %d iteration(s), %d accepted, %d CG step(s), Delta = %.3e, stop: %s\n",
             nodes_[a], S[a].fx + 0.0, S[a].gnorm, S[a].iteration, S[a].accepted, S[a].inner_total, S[a].Delta, names[S[a].status]);
    fflush(stdout);
  }
  for (int a : nodes) {
    res_[a].Gk = S[a].fx;
    res_[a].Gk_alt = S[a].fx - lin[a] + lin_alt[a];   // f(X | g_alt): only the linear term depends on g
    res_[a].tnt_status = S[a].status;
    res_[a].tnt_inner = S[a].inner_total;
  }
  return true;
}

}  // namespace dpgo
