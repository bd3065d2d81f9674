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
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "group.h"

namespace dpgo {

namespace {
// DPGO_CG_LAG=0: wait for every CG step's outcome before enqueueing the next (measurement hook)
int env_lag() {
  const char *e = getenv("DPGO_CG_LAG");
  return e ? atoi(e) : 1;
}
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

void Group::run_tnt(const std::vector<int> &nodes, double *X, const double *g, const double *g_alt, bool base_ready) {
  finish_update();
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
  auto quad_model = [&](const double *Y, bool from_base) {
    if (from_base) {
      apply_tcol(Y, T1_.p, nabla, 1, Y, nullptr, nullptr, grad);   // nabla and grad in one pass
      return;
    }
    launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, Y, false, g, nabla, nullptr, 0, nullptr, nullptr, 0);
    launch_tangent_rot(d_, st_, T_, cur_mask_, Y, nabla, grad);
  };
  auto precon = [&](const double *Y, const double *v, double *out) {
    if (!use_precon) {
      copy_rows(out, v, false, 0);
      return;
    }
    if (jacobi) launch_rot_rowscale(d_, st_, T_, cur_mask_, jacobi_.p, v, w1);
    else solve_rr(const_cast<double *>(v), w1, 1.0);   // w1.R = (G_RR + lambda I)^-1 v.R; the forward sweep only reads v
    launch_tangent_rot(d_, st_, T_, cur_mask_, Y, w1, out);
  };
  // gnorm, pgnorm (and rv0 = <grad, P grad>, the first CG scalar) of the nodes in `set` (mask == set).
  // with_f: f(X | g) in the same read-back, from the model gradient nabla = G X + g that is there anyway:
  //   f = <X, g> + 1/2 <X, G X> = 1/2 (<X, nabla> + <X, g>)      (DPGOProblem.cpp:180-205)
  // lin / lin_alt = <X, g> / <X, g_alt> let the caller re-base f on g_alt without another G X.
  std::vector<double> rv0(L, 0.0), lin(L, 0.0), lin_alt(L, 0.0);
  const double *ga = g_alt ? g_alt : g;
  auto norms = [&](const std::vector<int> &set, bool with_f) {
    {
      const double *pa[MAX_DOTS] = {grad, X, X, X}, *pb[MAX_DOTS] = {grad, nabla, g, ga};
      const int parts[MAX_DOTS] = {2, 0, 0, 0, 0, 0};
      launch_dots(d_, st_, T_, cur_mask_, with_f ? 4 : 1, pa, pb, parts, partials_.p, 0);
    }
    if (use_precon) {
      precon(X, grad, pg);
      const double *pa[MAX_DOTS] = {pg, grad}, *pb[MAX_DOTS] = {pg, pg};
      launch_dots(d_, st_, T_, cur_mask_, 2, pa, pb, P2, partials_.p, MAX_DOTS);
    }
    fetch(MAX_DOTS + 2, false);
    for (int a : set) {
      S[a].gnorm = S[a].pgnorm = std::sqrt(scal(a, 0));
      rv0[a] = scal(a, 0);
      if (use_precon) {
        S[a].pgnorm = std::sqrt(scal(a, MAX_DOTS));
        rv0[a] = scal(a, MAX_DOTS + 1);
      }
      if (with_f) {
        S[a].fx = 0.5 * (scal(a, 1) + scal(a, 2)) + res_[a].f;
        lin[a] = scal(a, 2);
        lin_alt[a] = scal(a, 3);
      }
    }
  };

  const double sqrt_eps = std::sqrt(std::numeric_limits<double>::epsilon());
  const double eta1 = .05, eta2 = .9, alpha1 = .25, alpha2 = 2.5, Delta_tol = 1e-6;   // TNT.h:83-97,129
  set_mask(nodes);
  quad_model(X, base_ready);
  norms(nodes, true);

  for (;;) {
    // ---- nodes that start another trust-region iteration (TNT.h:446-484)
    std::vector<int> A;
    for (int a : nodes) {
      NodeTnt &s = S[a];
      if (!s.active) continue;
      if (!(s.iteration < o.max_iterations && s.accepted < o.max_iterations_accepted)) { s.active = false; continue; }
      if (s.gnorm < o.grad_norm_tol) { s.status = ST_GRADIENT; s.active = false; continue; }
      if (s.pgnorm < o.preconditioned_grad_norm_tol) { s.status = ST_PRECON_GRADIENT; s.active = false; continue; }
      A.push_back(a);
    }
    if (A.empty()) break;
    // ---- STPCG (IterativeSolvers.h:207-426).  The scalar recurrences (alpha, beta, the boundary / negative
    // curvature / kernel tests, the stopping test) run on the device (k_cg_scal); the vector kernels take their
    // step lengths and the set of still-iterating nodes from device memory, so a whole CG step is enqueued
    // without a host round trip.  The host only polls the summary (live, |h|_M, iterations) of a step it enqueued
    // earlier: step i+1 is already queued when the outcome of step i arrives; once every node has stopped, the
    // kernels of the surplus step find an empty device mask and return at once.
    set_mask(A);
    const NodeBits bitsA = cur_mask_.v;
    // p_0 = -v_0, v_0 = P(grad) (already there from the preconditioned gradient norm);
    // s_0 = 0, H s_0 = 0, r_0 = grad are not materialised: the first step of the run takes them as given
    launch_cg_init(d_, st_, T_, cur_mask_, grad, use_precon ? pg : grad, nullptr, nullptr, nullptr, nullptr, pk);
    bool first_step = true;
    {
      CgStart cs;
      for (int a = 0; a < L; a++) cs.rv[a] = cs.Delta[a] = cs.target[a] = 0.0;
      for (int a : A) {
        cs.rv[a] = rv0[a];   // <r_0, v_0> = <grad, P grad>, read back together with the norms
        cs.Delta[a] = S[a].Delta;
        const double r0 = std::sqrt(rv0[a]);
        cs.target[a] = r0 * std::min(o.STPCG_kappa, std::pow(r0, o.STPCG_theta));
      }
      launch_cg_begin(st_, L, bitsA, cs, o.max_tCG_iterations, cg_.p, dmask_.p);
    }
    const NodeMask mA{bitsA, dmask_.p}, mB{bitsA, dmask_.p + 1};
    // first half of a step: H p and its four scalars, then the step-length logic (:296-362)
    auto stepA = [&]() {
      cur_mask_ = mA;
      launch_bsr(d_, st_, T_, false, cur_mask_, G_.dev, pk, true, nullptr, w1, nullptr, 0, nullptr, nullptr, 0);   // G [0 ; p.R]
      solve_tt(w1, w3, -1.0);
      apply_tcol(w3, w1, nullptr, 2, X, nabla, pk, Hp, first_step ? grad : rk, partials_.p);   // Hp and <p,Hp>, <Hp,Hp>, <p,p>, <p,r>
      launch_cg_scal(st_, T_, L, 0, partials_.p, cg_.p, dmask_.p, h_scal_, reduce_arrived_.p, h_flag_, ++fetch_seq_);
      // s += c1 p, H s += c1 H p for every node of the step (a node that stops here takes its boundary step), r += alpha H p
      // for those that go on
      // (the first step runs for every node of A, live or not: a node that stops before its first step has c1 = 0 and
      // gets its s = H s = 0 written here)
      launch_cg_step(d_, st_, T_, first_step ? NodeMask{bitsA, nullptr} : mA, NodeCoefs(), pk, Hp, sk, hh, rk, cg_.p,
                     first_step ? grad : nullptr);
      first_step = false;
      return fetch_seq_;
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
      launch_cg_scal(st_, T_, L, 1, partials_.p, cg_.p, dmask_.p, h_scal_, reduce_arrived_.p, h_flag_, ++fetch_seq_);
      launch_cg_dir(d_, st_, T_, cur_mask_, cg_.p, vk, pk);
      return fetch_seq_;
    };
    auto any_live = [&]() {
      for (int a : A)
        if (scal(a, 0) != 0.0) return true;
      return false;
    };
    static const int lag = env_lag();
    // the first step is awaited at once: in the early regime every node ends it on the trust-region boundary
    wait_flag(stepA());
    if (any_live()) {
      unsigned long long seqB = stepB();
      for (;;) {
        if (!lag) {
          wait_flag(seqB);
          if (!any_live()) break;
        }
        stepA();
        const unsigned long long next = stepB();
        if (lag) {
          wait_flag(seqB);   // the outcome of the step before the one just enqueued
          if (!any_live()) break;
        }
        seqB = next;
      }
    }
    for (int a : A) {
      S[a].h_M_norm = scal(a, 1);
      S[a].cg_it = (int)scal(a, 2);
      S[a].cg = false;
    }
    for (int a : A) S[a].inner_total += S[a].cg_it;
    // ---- trial point (TNT.h:505-536)
    set_mask(A);
    launch_retract_rot(d_, st_, T_, cur_mask_, X, sk, xprop);
    recover_translations(xprop, g);
    apply_tcol(xprop, T1_.p, nprop);          // nprop = G xprop + g: gives f(xprop) and, if accepted, the next model
    {
      const double *pa[MAX_DOTS] = {sk, grad, sk, xprop, xprop, xprop}, *pb[MAX_DOTS] = {sk, sk, hh, g, ga, nprop};
      const int parts[MAX_DOTS] = {2, 2, 2, 0, 0, 0};
      launch_dots(d_, st_, T_, cur_mask_, 6, pa, pb, parts, partials_.p, 0);
    }
    fetch(MAX_DOTS, false);
    std::vector<int> acc, requad;
    for (int a : A) {
      NodeTnt &s = S[a];
      const double fx_prop = 0.5 * (scal(a, 5) + scal(a, 3)) + res_[a].f;
      const double h_norm = std::sqrt(scal(a, 0));
      const double dm = -scal(a, 1) - 0.5 * scal(a, 2);
      const double df = s.fx - fx_prop;
      const double rel_dec = df / (sqrt_eps + std::fabs(s.fx));
      const double rho = df / dm;
      const bool ok = (!std::isnan(rho)) && rho > eta1;
      s.accepted += ok;
      bool stop = false;
      if (ok) {
        acc.push_back(a);
        s.fx = fx_prop;
        lin[a] = scal(a, 3);
        lin_alt[a] = scal(a, 4);
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
    }
    if (!acc.empty()) {
      set_mask(acc);
      copy_rows(X, xprop, false, 0);
    }
    if (!requad.empty()) {
      set_mask(requad);
      copy_rows(nabla, nprop, false, 0);   // the model gradient at the accepted point
      launch_tangent_rot(d_, st_, T_, cur_mask_, X, nabla, grad);
      norms(requad, false);
    }
  }
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
}

}  // namespace dpgo
