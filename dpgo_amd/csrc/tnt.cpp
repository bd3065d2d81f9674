// Riemannian truncated-Newton trust-region refinement of one node on device
// vectors.
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
// The control flow (a handful of scalars per CG step) runs on the host; every
// vector lives in HBM and is touched only by kernels, restricted to the node by
// the device mask.
#include <cmath>
#include <cstdio>
#include <limits>

#include "group.h"

namespace dpgo {

namespace {
enum { ST_GRADIENT = 0, ST_PRECON_GRADIENT, ST_REL_DECREASE, ST_STEPSIZE, ST_TRUST_REGION, ST_ITER_LIMIT };
}

void Group::run_tnt(int a, double *X, const double *g) {
  const Options &o = opt_;
  NodeResults &res = res_[a];
  set_mask({a});
  const bool use_precon = (o.preconditioner == 1) && Lrr_.F.n > 0;
  // work vectors (own rows; only node a's rows are touched)
  double *nabla = tmp_[0].p, *grad = tmp_[1].p, *sk = tmp_[2].p, *rk = tmp_[3].p, *vk = tmp_[4].p, *pk = tmp_[5].p,
         *Hp = tmp_[6].p, *xprop = tmp_[7].p, *w1 = tmp_[8].p, *w2 = tmp_[9].p, *pg = tmp_[10].p, *hh = tmp_[11].p;
  const double fconst = res.f;

  auto dots = [&](std::initializer_list<std::pair<const double *, const double *>> prs, double *out) {
    int s = 0;
    for (const auto &pr : prs) launch_dot(d_, st_, T_, false, mask_.p, pr.first, pr.second, 2, partials_.p, s++);
    fetch(s, false);
    for (int k = 0; k < s; k++) out[k] = scal(a, k);
  };
  auto fval = [&](const double *Y) {
    eval_G(Y, g, 0);
    fetch(1, false);
    return scal(a, 0) + fconst;
  };
  auto quad_model = [&](const double *Y) {   // nabla, grad at Y
    launch_bsr(d_, st_, T_, false, mask_.p, G_.dev, Y, false, g, nabla, nullptr, 0, nullptr, nullptr, 0);
    launch_tangent_rot(d_, st_, T_, mask_.p, Y, nabla, grad);
  };
  auto hess = [&](const double *Y, const double *v, double *out) {
    launch_bsr(d_, st_, T_, false, mask_.p, G_.dev, v, true, nullptr, w1, nullptr, 0, nullptr, nullptr, 0);
    solve_tt(w1, -1.0);                       // w1.t = tdot
    copy_rows(w1, v, false, 2);               // w1 = [tdot ; Rdot]
    launch_bsr(d_, st_, T_, false, mask_.p, G_.dev, w1, false, nullptr, w2, nullptr, 0, nullptr, nullptr, 0);
    launch_hess_epilogue(d_, st_, T_, mask_.p, Y, w2, nabla, v, out);
  };
  auto precon = [&](const double *Y, const double *v, double *out) {
    if (!use_precon) {
      copy_rows(out, v, false, 0);
      return;
    }
    copy_rows(w1, v, false, 0);
    solve_rr(w1, 1.0);
    launch_tangent_rot(d_, st_, T_, mask_.p, Y, w1, out);
  };
  auto retract = [&](const double *Y, const double *v, double *out) {
    launch_retract_rot(d_, st_, T_, mask_.p, Y, v, out);
    recover_translations(out, g);
  };
  auto axpby = [&](double al, const double *x, double be, const double *y, double *out) {
    launch_axpby(d_, st_, T_, false, mask_.p, al, x, be, y, out, 0);
  };

  const double sqrt_eps = std::sqrt(std::numeric_limits<double>::epsilon());
  int status = ST_ITER_LIMIT;
  double fx = fval(X);
  quad_model(X);
  double sc[4];
  dots({{grad, grad}}, sc);
  double gnorm = std::sqrt(sc[0]), pgnorm = gnorm;
  if (use_precon) {
    precon(X, grad, pg);
    dots({{pg, pg}}, sc);
    pgnorm = std::sqrt(sc[0]);
  }
  double Delta = 1.0;   // TNTParams::Delta0 (TNT.h:81)
  const double eta1 = .05, eta2 = .9, alpha1 = .25, alpha2 = 2.5, Delta_tol = 1e-6;
  int accepted = 0, inner_total = 0;
  for (int iteration = 0; iteration < o.max_iterations && accepted < o.max_iterations_accepted; ++iteration) {
    if (gnorm < o.grad_norm_tol) { status = ST_GRADIENT; break; }
    if (pgnorm < o.preconditioned_grad_norm_tol) { status = ST_PRECON_GRADIENT; break; }
    // ---- STPCG (IterativeSolvers.h:207-426)
    double h_M_norm = 0;
    {
      axpby(0.0, grad, 0.0, nullptr, sk);
      copy_rows(rk, grad, false, 0);
      if (use_precon) precon(X, rk, vk); else copy_rows(vk, rk, false, 0);
      axpby(-1.0, vk, 0.0, nullptr, pk);
      double sk_M_pk = 0, sk_M_2 = 0;
      dots({{rk, vk}}, sc);
      double rv = sc[0];
      double pk_M_2 = rv;
      const double Delta_2 = Delta * Delta;
      const double r0 = std::sqrt(rv);
      const double target = r0 * std::min(o.STPCG_kappa, std::pow(r0, o.STPCG_theta));
      bool done = false;
      int it = 0;
      for (; it < o.max_tCG_iterations; ++it) {
        if (std::sqrt(rv) <= target) break;
        hess(X, pk, Hp);
        dots({{pk, Hp}, {Hp, Hp}, {pk, pk}, {pk, rk}}, sc);
        const double kappa_k = sc[0];
        if (std::sqrt(sc[1]) / std::sqrt(sc[2]) < 1e-8) {   // :305-338
          double sgn = 1.0;
          if (sc[3] < 0) { sgn = -1.0; sk_M_pk = -sk_M_pk; }
          const double sigma = (-sk_M_pk + std::sqrt(sk_M_pk * sk_M_pk + pk_M_2 * (Delta_2 - sk_M_2))) / pk_M_2;
          axpby(1.0, sk, sgn * sigma, pk, sk);
          h_M_norm = Delta;
          done = true;
          break;
        }
        const double alpha = rv / kappa_k;
        const double skp1 = sk_M_2 + 2 * alpha * sk_M_pk + alpha * alpha * pk_M_2;
        if (kappa_k <= 0 || skp1 > Delta_2) {               // :347-362
          const double sigma = (-sk_M_pk + std::sqrt(sk_M_pk * sk_M_pk + pk_M_2 * (Delta_2 - sk_M_2))) / pk_M_2;
          axpby(1.0, sk, sigma, pk, sk);
          h_M_norm = Delta;
          done = true;
          break;
        }
        axpby(1.0, sk, alpha, pk, sk);
        axpby(1.0, rk, alpha, Hp, rk);
        if (use_precon) precon(X, rk, vk); else copy_rows(vk, rk, false, 0);
        dots({{rk, vk}}, sc);
        const double rk_vk = sc[0];
        const double beta = rk_vk / (alpha * kappa_k);
        sk_M_2 = skp1;
        sk_M_pk = beta * (sk_M_pk + alpha * pk_M_2);
        pk_M_2 = rk_vk + beta * beta * pk_M_2;
        axpby(-1.0, vk, beta, pk, pk);
        rv = rk_vk;
      }
      if (!done) h_M_norm = std::sqrt(sk_M_2);
      inner_total += it;
    }
    // ---- trial point (TNT.h:505-536)
    retract(X, sk, xprop);
    const double fx_prop = fval(xprop);
    hess(X, sk, hh);
    dots({{sk, sk}, {grad, sk}, {sk, hh}}, sc);
    const double h_norm = std::sqrt(sc[0]);
    const double dm = -sc[1] - 0.5 * sc[2];
    const double df = fx - fx_prop;
    const double rel_dec = df / (sqrt_eps + std::fabs(fx));
    const double rho = df / dm;
    const bool step_accepted = (!std::isnan(rho)) && rho > eta1;
    accepted += step_accepted;
    if (step_accepted) {
      copy_rows(X, xprop, false, 0);
      fx = fx_prop;
      if (rel_dec < o.rel_func_decrease_tol) { status = ST_REL_DECREASE; break; }
      if (h_norm < o.stepsize_tol) { status = ST_STEPSIZE; break; }
      quad_model(X);
      dots({{grad, grad}}, sc);
      gnorm = std::sqrt(sc[0]);
      pgnorm = gnorm;
      if (use_precon) {
        precon(X, grad, pg);
        dots({{pg, pg}}, sc);
        pgnorm = std::sqrt(sc[0]);
      }
    }
    if ((!std::isnan(rho)) && rho >= eta2) {
      Delta = std::max(alpha2 * h_M_norm, Delta);
    } else if (std::isnan(rho) || rho < eta1) {
      Delta = alpha1 * h_M_norm;
      if (Delta < Delta_tol) { status = ST_TRUST_REGION; break; }
    }
  }
  res.Gk = fx;
  res.tnt_status = status;
  res.tnt_inner = inner_total;
}

}  // namespace dpgo
