"""Oracle: Steihaug-Toint preconditioned CG and the Riemannian truncated-Newton
trust-region method.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  * STPCG  C++/Optimization/include/Optimization/LinearAlgebra/IterativeSolvers.h:166-426
    (no linear constraints: the At / lambda branch of :236-252 is not used by DPGO)
  * TNT    C++/Optimization/include/Optimization/Riemannian/TNT.h:242-693
including the extra ``max_iterations_accepted`` stop (TNT.h:446-449,
Base/Concepts.h:47-48).  PINNED against the reference headers compiled as
oracle/_ref/tnt_ref (tests/test_oracle_tnt.py).
"""
from __future__ import annotations

import math

import numpy as np


def _sqrt(x):
    """std::sqrt semantics: NaN for a negative argument (Python's math.sqrt raises instead), so that a
    rounding-level negative <r, P r> behaves as it does in the reference (the comparison is false)."""
    return math.sqrt(x) if x >= 0 else float("nan")


class TNTParams:
    def __init__(self):
        # SmoothOptimizerParams / OptimizerParams defaults (Base/Concepts.h:42-63)
        self.max_iterations = 100
        self.max_iterations_accepted = 100
        self.gradient_tolerance = 1e-6
        self.relative_decrease_tolerance = 1e-6
        self.stepsize_tolerance = 1e-6
        # TNTParams defaults (TNT.h:76-130)
        self.Delta0 = 1.0
        self.eta1 = 0.05
        self.eta2 = 0.9
        self.alpha1 = 0.25
        self.alpha2 = 2.5
        self.max_TPCG_iterations = 1000
        self.kappa_fgr = 0.1
        self.theta = 0.5
        self.preconditioned_gradient_tolerance = 1e-6
        self.Delta_tolerance = 1e-6


def stpcg(g, H, inner, Delta, max_iterations=1000, kappa_fgr=0.1, theta=0.5,
          P=None, epsilon=1e-8, trace=None):
    """IterativeSolvers.h:166-426.  Returns (s, update_step_M_norm, num_iterations)."""
    s_k = 0 * g
    r_k = g.copy()
    v_k = r_k if P is None else P(r_k)
    p_k = -v_k
    sk_M_pk = 0.0
    sk_M_2 = 0.0
    pk_M_2 = inner(r_k, v_k)
    Delta_2 = Delta * Delta
    r0_norm = _sqrt(inner(r_k, v_k))
    target = r0_norm * min(kappa_fgr, r0_norm ** theta)
    it = 0
    while it < max_iterations:
        if _sqrt(inner(r_k, v_k)) <= target:                    # :290
            break
        Hp = H(p_k)
        kappa_k = inner(p_k, Hp)
        if _sqrt(inner(Hp, Hp)) / _sqrt(inner(p_k, p_k)) < epsilon:   # :305-338
            if inner(p_k, r_k) < 0:
                p_k = -p_k
                sk_M_pk = -sk_M_pk
            sigma = (-sk_M_pk + _sqrt(sk_M_pk * sk_M_pk + pk_M_2 * (Delta_2 - sk_M_2))) / pk_M_2
            return s_k + sigma * p_k, Delta, it
        alpha = inner(r_k, v_k) / kappa_k
        skp1_M_2 = sk_M_2 + 2 * alpha * sk_M_pk + alpha * alpha * pk_M_2
        if kappa_k <= 0 or skp1_M_2 > Delta_2:                      # :347-362
            sigma = (-sk_M_pk + _sqrt(sk_M_pk * sk_M_pk + pk_M_2 * (Delta_2 - sk_M_2))) / pk_M_2
            return s_k + sigma * p_k, Delta, it
        s_k = s_k + alpha * p_k
        r_k = r_k + alpha * Hp
        v_k = r_k if P is None else P(r_k)
        rk_vk = inner(r_k, v_k)
        beta = rk_vk / (alpha * kappa_k)
        sk_M_2 = skp1_M_2
        sk_M_pk = beta * (sk_M_pk + alpha * pk_M_2)
        pk_M_2 = rk_vk + beta * beta * pk_M_2
        p_k = -v_k + beta * p_k
        if trace is not None:
            trace.append((alpha, beta))
        it += 1
    return s_k, _sqrt(sk_M_2), it


def tnt(f, QM, metric, retract, x0, precon=None, params=None, log=None):
    """TNT.h:242-693.

    f(x) -> scalar; QM(x) -> (grad, Hess) with Hess(x, v) -> tangent;
    metric(x, v1, v2) -> scalar; retract(x, v) -> point;
    precon(x, v) -> tangent or None.  Returns dict(x, f, status, ...)."""
    p = params or TNTParams()
    sqrt_eps = math.sqrt(np.finfo(np.float64).eps)
    status = "IterationLimit"
    x = x0
    fx = f(x)
    grad, Hess = QM(x)
    gnorm = _sqrt(metric(x, grad, grad))
    if precon is not None:
        pg = precon(x, grad)
        pgnorm = _sqrt(metric(x, pg, pg))
    else:
        pgnorm = gnorm
    Delta = p.Delta0
    iteration = 0
    accepted = 0
    inner_its = []
    while iteration < p.max_iterations and accepted < p.max_iterations_accepted:
        if gnorm < p.gradient_tolerance:
            status = "Gradient"
            break
        if pgnorm < p.preconditioned_gradient_tolerance:
            status = "PreconditionedGradient"
            break
        xc, Hc = x, Hess
        h, h_M_norm, nin = stpcg(
            grad, lambda v: Hc(xc, v), lambda a, b: metric(xc, a, b), Delta,
            p.max_TPCG_iterations, p.kappa_fgr, p.theta,
            (lambda v: precon(xc, v)) if precon is not None else None)
        inner_its.append(nin)
        h_norm = _sqrt(metric(x, h, h))
        x_prop = retract(x, h)
        fx_prop = f(x_prop)
        dm = -metric(x, grad, h) - 0.5 * metric(x, h, Hess(x, h))
        df = fx - fx_prop
        rel_dec = df / (sqrt_eps + abs(fx))
        with np.errstate(divide="ignore", invalid="ignore"):
            rho = float(np.float64(df) / np.float64(dm))
        step_accepted = (not math.isnan(rho)) and rho > p.eta1
        accepted += int(step_accepted)
        if log is not None:
            log.append(dict(iteration=iteration, fx=fx, gnorm=gnorm, pgnorm=pgnorm, Delta=Delta,
                            inner=nin, h_norm=h_norm, h_M_norm=h_M_norm, df=df, rho=rho,
                            accepted=step_accepted))
        if step_accepted:
            x = x_prop
            fx = fx_prop
            if rel_dec < p.relative_decrease_tolerance:
                status = "RelativeDecrease"
                break
            if h_norm < p.stepsize_tolerance:
                status = "Stepsize"
                break
            grad, Hess = QM(x)
            gnorm = _sqrt(metric(x, grad, grad))
            if precon is not None:
                pg = precon(x, grad)
                pgnorm = _sqrt(metric(x, pg, pg))
            else:
                pgnorm = gnorm
        if (not math.isnan(rho)) and rho >= p.eta2:
            Delta = max(p.alpha2 * h_M_norm, Delta)
        elif math.isnan(rho) or rho < p.eta1:
            Delta = p.alpha1 * h_M_norm
            if Delta < p.Delta_tolerance:
                status = "TrustRegion"
                break
        iteration += 1
    return dict(x=x, f=fx, status=status, gradfx_norm=gnorm, preconditioned_gradfx_norm=pgnorm,
                Delta=Delta, inner_iterations=inner_its)
