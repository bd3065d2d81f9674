"""Oracle: the per-node AMM-PGO# / MM-PGO optimizer.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates C++/DPGO/src/DPGOHash.cpp (update :84-228, amm_pgo :230-444,
mm_pgo :446-581, iterate :583-628) and the halo copy of
C++/DPGO/include/DPGO/DPGOHash.h:28-86, with the solver options of
C++/DPGO/include/DPGO/DPGO_types.h:78-201 and the driver overrides of
C++/examples/dist_pgo.cpp:103-120.  History vectors are kept as two-deep
ring buffers (the reference keeps them unbounded, DPGOHash.cpp:99-106).
"""
from __future__ import annotations

import math

import numpy as np

from . import tnt as tnt_mod
from .problem import DPGOProblem, LOSS_NONE

SCHEME_MM, SCHEME_AMM = 0, 1


def _div(a, b):
    """IEEE division as in C++ (inf / nan instead of ZeroDivisionError): a node whose objective is exactly
    zero is still handled the way the reference handles it."""
    if b == 0.0:
        return float("nan") if a == 0.0 or a != a else math.copysign(float("inf"), a)
    return a / b


class Options:
    """DPGO::Options (DPGO_types.h:78-201) with dist_pgo.cpp:103-120 applied by
    ``Options.driver()``."""

    def __init__(self):
        self.scheme = SCHEME_AMM
        self.regularizer = 1e-10
        self.accepted_delta = 5e-4
        self.eta = [5e-4, 2.5e-2]
        self.psi = 1e-10
        self.phi = 1e-6
        self.max_soft_restart_hits = [10, 25]
        self.oscillation_cnt_period = 15
        self.max_oscillations = 12
        self.loss = LOSS_NONE
        self.loss_reg = 1.0
        self.rescale = 1                                          # Rescale::Dynamic (DPGO_types.h:128)
        self.max_rescale_count = 5                                # DPGO_types.h:131
        self.grad_norm_tol = 5e-3
        self.rel_func_decrease_tol = 1e-6
        self.stepsize_tol = 1e-4
        self.max_iterations = 10
        self.max_iterations_accepted = 1
        self.preconditioner = 3                                   # Preconditioner::RegularizedCholesky (DPGO_types.h:155)
        self.reg_Cholesky_precon_max_condition_number = 1e6
        self.preconditioned_grad_norm_tol = 1e-4
        self.max_tCG_iterations = 10000
        self.STPCG_kappa = 0.05
        self.STPCG_theta = 0.9

    @staticmethod
    def driver(loss=LOSS_NONE, accelerated=True):
        o = Options()
        o.loss = loss
        o.rescale = 0                                             # Rescale::Static (dist_pgo.cpp:105)
        o.loss_reg = 0.25
        o.scheme = SCHEME_AMM if accelerated else SCHEME_MM
        o.grad_norm_tol = 1e-3
        o.preconditioned_grad_norm_tol = 1e-4
        o.regularizer = 1e-11
        return o


class Results:
    """The fields of DPGOResult (DPGO_types.h:204-322) the state machine uses."""

    def __init__(self):
        self.updated = True
        self.Xk = None
        self.Xak = None
        self.Xakh = None
        self.gradFnorm = 0.0
        self.fobjE = 0.0
        self.DfobjE = None
        self.Fk = [0.0, 0.0]
        self.Gk = 0.0
        self.X = [None, None]        # X[iter], X[iter-1]
        self.g = [None, None]
        self.Dfobj = [None, None]
        self.fobj = [0.0, 0.0]
        self.f = 0.0
        self.gamma = 0.0
        self.s = [1.0, 1.0]          # s[iter], s[iter+1]
        self.soft_restart_hits = [0, 0]
        self.oscillations = []
        self.num_oscillations = 0
        self.iters = 0
        self.refined = False         # trace only
        self.tnt_status = ""         # trace only


class DPGOHash:
    def __init__(self, node, measurements, options):
        self.options = options
        self.problem = DPGOProblem(
            node, measurements, options.regularizer, options.loss,
            options.reg_Cholesky_precon_max_condition_number, options.loss_reg,
            preconditioner=getattr(options, "preconditioner", 3),
            dynamic=(getattr(options, "rescale", 0) == 1))                   # DPGOHash.cpp:16
        self.results = Results()

    # DPGOHash.cpp:20-43
    def initialize(self, X):
        p = self.problem
        d = p.d
        assert X.shape == ((d + 1) * (p.n[0] + p.n[1]), d)
        r = self.results = Results()
        r.Xk = X.copy()
        r.Xak = r.Xk[:(d + 1) * p.n[0]].copy()
        r.gamma = 0.0
        r.updated = False
        return 0

    # DPGOHash.h:28-86
    def communicate(self, pgos):
        p = self.problem
        d, n, s = p.d, p.n, p.s
        Xk = self.results.Xk
        for beta, poses in p.info.index.items():
            if beta == p.node:
                continue
            src = pgos[beta].results.Xk
            nb0 = pgos[beta].problem.n[0]
            for j, (_, k) in poses.items():
                Xk[s[1] * (d + 1) + k] = src[j]
                r0 = s[1] * (d + 1) + n[1] + k * d
                Xk[r0:r0 + d] = src[nb0 + j * d: nb0 + j * d + d]
        return 0

    # DPGOHash.cpp:45-82
    def receive(self, msg):
        """msg: {beta: ((d+1) |recv[beta]|) x d matrix [t rows ; R rows]}; clears `updated`."""
        p, r = self.problem, self.results
        d, n, s = p.d, p.n, p.s
        for b, M in msg.items():
            if b not in p.info.recv:
                return -1
            r.updated = False
            poses = p.info.recv[b]
            k = len(poses)
            assert M.shape == ((d + 1) * k, d)
            i = next(iter(poses.values()))[1]
            r.Xk[s[1] * (d + 1) + i: s[1] * (d + 1) + i + k] = M[:k]
            r0 = s[1] * (d + 1) + n[1] + i * d
            r.Xk[r0:r0 + k * d] = M[k:]
        return 0

    # DPGOHash.cpp:84-228
    def update(self):
        r, p, o = self.results, self.problem, self.options
        if r.updated:
            return 0
        it = r.iters
        # history: index 0 = current iteration, 1 = previous.  The reference writes X[iter], g[iter], ... in place
        # (DPGOHash.cpp:99-106), so a second update() at the same iteration (after receive()) keeps X[iter-1],
        # g[iter-1], fobj[iter-1] and s[iter]; the counters below are simply run again, as in the reference.
        repeat = getattr(r, "hist_iter", -1) == it
        r.hist_iter = it
        h = 1 if repeat else 0
        r.X = [r.Xk.copy(), r.X[h]]
        prev_fobj = r.fobj[h]
        if p.trivial:
            if it == 0:
                g, f = p.evaluate_none_g_and_f0(r.X[0])
                fobj = p.evaluate_G(r.Xak, g, f)
            else:
                g, f, fobj = p.evaluate_none_g_and_f(r.X[0], r.X[1], r.Gk)
            Dfobj = None
        else:
            if p.dynamic:        # Rescale::Dynamic (DPGOHash.cpp:131-144)
                rc = getattr(r, "rescale_count", 0)
                if it == 0:
                    g, f, Dfobj, fobj, r.DfobjE, r.fobjE, rc = p.evaluate_g_and_f0_rescale(r.X[0], rc, o.max_rescale_count)
                else:
                    g, f, Dfobj, fobj, r.DfobjE, r.fobjE, rc = p.evaluate_g_and_f_rescale(
                        r.X[0], r.X[1], r.Gk, r.DfobjE, r.fobjE, rc, o.max_rescale_count)
                r.rescale_count = rc
            elif it == 0:
                g, f, Dfobj, fobj, r.DfobjE, r.fobjE = p.evaluate_g_and_f0(r.X[0])
            else:
                g, f, Dfobj, fobj, r.DfobjE, r.fobjE = p.evaluate_g_and_f(
                    r.X[0], r.X[1], r.Gk, r.DfobjE, r.fobjE)
        r.g = [g, r.g[h]]
        r.f = f
        r.fobj = [fobj, prev_fobj]
        if it == 0:
            r.Fk = [fobj, fobj]
            r.Gk = fobj
        if p.trivial:
            Dfobj, gradF = p.full_Riemannian_gradient_G(r.Xak, g)
        else:
            gradF = p.full_tangent_space_projection(r.Xak, Dfobj)
        r.Dfobj = [Dfobj, r.Dfobj[h]]
        r.gradFnorm = float(np.linalg.norm(gradF))
        if o.scheme == SCHEME_AMM:
            if it == 0:
                r.s = [1.0, 1.0]
                r.oscillations = [1] if not repeat else r.oscillations + [1]   # push_back (DPGOHash.cpp:168-171)
            elif not repeat:
                r.s = [r.s[1], 0.0]
            s0 = r.s[0]
            s1 = 0.5 + 0.5 * math.sqrt(4.0 * s0 * s0 + 1.0)
            r.s[1] = s1
            r.gamma = (s0 - 1) / s1
            if fobj <= r.Fk[1]:
                r.soft_restart_hits[0] = r.soft_restart_hits[0] - 2 if r.soft_restart_hits[0] > 2 else 0
            else:
                r.soft_restart_hits[0] += 1
            if it > 0:
                if fobj <= prev_fobj:
                    r.soft_restart_hits[1] = 0
                    r.oscillations.append(1)
                else:
                    r.soft_restart_hits[1] += 1
                    r.oscillations.append(0)
                r.num_oscillations += int(r.oscillations[it] != r.oscillations[it - 1])
            if it > o.oscillation_cnt_period:
                k = it - o.oscillation_cnt_period
                r.num_oscillations -= int(r.oscillations[k] != r.oscillations[k - 1])
            r.Fk[0] = r.Fk[0] * (1 - o.eta[0]) + fobj * o.eta[0]
            r.Fk[1] = max(fobj, r.Fk[1] * (1 - o.eta[1]) + fobj * o.eta[1])
        else:
            r.Fk = [fobj, fobj]
        r.updated = True
        return 0

    def _tnt(self, x0, g):
        """The TNT call of DPGOHash.cpp:270-349, 374-381."""
        p, o, r = self.problem, self.options, self.results
        n0 = p.n[0]
        f = r.f
        cache = {}

        def Fobj(Y):
            return p.evaluate_G(Y, g, f)

        def QM(Y):
            nabla = p.reduced_Euclidean_gradient_G(Y, g)
            cache["nabla"] = nabla
            grad = p.reduced_tangent_space_projection(Y, nabla)

            def Hess(Yc, Ydot):
                return p.hessian_vector_product(Yc, cache["nabla"], Ydot)
            return grad, Hess

        def metric(Y, V1, V2):
            return float(np.sum(V1 * V2))          # tr(V1 V2^T), DPGOHash.cpp:307-310

        def retract(Y, Ydot):
            return p.retract(Y, Ydot, g)

        precon = (lambda Y, V: p.precondition(Y, V)) if (p.precon is not None or p.jacobi is not None) else None
        prm = tnt_mod.TNTParams()
        prm.gradient_tolerance = o.grad_norm_tol
        prm.preconditioned_gradient_tolerance = o.preconditioned_grad_norm_tol
        prm.relative_decrease_tolerance = o.rel_func_decrease_tol
        prm.stepsize_tolerance = o.stepsize_tol
        prm.max_iterations = o.max_iterations
        prm.max_iterations_accepted = o.max_iterations_accepted
        prm.max_TPCG_iterations = o.max_tCG_iterations
        prm.kappa_fgr = o.STPCG_kappa
        prm.theta = o.STPCG_theta
        res = tnt_mod.tnt(Fobj, QM, metric, retract, x0, precon, prm)
        r.tnt_status = res["status"]
        return res

    # DPGOHash.cpp:230-444
    def amm_pgo(self):
        r, p, o = self.results, self.problem, self.options
        n0, d = p.n[0], p.d
        it = r.iters
        if it == 0:
            Y = r.Xk
            g = r.g[0]
            Df = r.Dfobj[0]
        else:
            Y = r.X[0] + r.gamma * (r.X[0] - r.X[1])
            if p.trivial:
                g = r.g[0] + r.gamma * (r.g[0] - r.g[1])
                Df = r.Dfobj[0] + r.gamma * (r.Dfobj[0] - r.Dfobj[1])
            else:
                g, Df = p.evaluate_g_and_Df(Y)
        f = r.f
        refined = ((_div(r.gradFnorm * r.gradFnorm, r.fobj[0]) > o.accepted_delta)
                   or (r.num_oscillations >= o.max_oscillations)) \
            and o.max_iterations > 0 and o.max_iterations_accepted > 0
        r.refined = refined
        Fk = r.Fk
        r.Xakh = p.proximal(Y, Df)
        Gkh = p.evaluate_G(r.Xakh, r.g[0], f)
        minG = Fk[0] - o.psi * float(np.sum((r.Xakh - r.Xak) ** 2))
        R = r.Xakh[n0:].copy()
        t = p.recover_translations(R, g)
        r.Xak = np.vstack([t, R])
        if refined:
            r.Xak = self._tnt(r.Xak, g)["x"]
        r.Gk = p.evaluate_G(r.Xak, r.g[0], f)
        if Gkh > minG:                                           # :386-389
            r.Xakh = p.proximal(r.Xk, r.Dfobj[0])
            Gkh = p.evaluate_G(r.Xakh, r.g[0], f)
        hard_restart = r.Gk > Fk[0]
        soft_restart = (r.Gk > Fk[1] and r.soft_restart_hits[0] >= o.max_soft_restart_hits[0]) or \
                       (r.Gk > r.fobj[0] and r.soft_restart_hits[1] > o.max_soft_restart_hits[1])
        if hard_restart or soft_restart:                         # :402-432
            g = r.g[0]
            if Gkh <= r.fobj[0]:
                r.Xak = r.Xakh.copy()
            else:
                r.Xak = p.proximal(r.Xk, r.Dfobj[0])
            R = r.Xak[n0:]
            t = p.recover_translations(R, r.g[0])
            r.Xak = np.vstack([t, R])
            if refined:
                res = self._tnt(r.Xak, g)
                r.Xak = res["x"]
                r.Gk = res["f"]
            else:
                r.Gk = p.evaluate_G(r.Xak, r.g[0], f)
            if hard_restart:
                r.s[1] = max(0.5 * r.s[1], 1.0)
            r.soft_restart_hits[0] //= 3
            r.soft_restart_hits[1] = 0
        if (Fk[0] - r.Gk) < o.phi * (Fk[0] - Gkh):               # :434-441
            R = r.Xakh[n0:].copy()
            t = p.recover_translations(R, g)
            r.Xak = np.vstack([t, R])
            r.Gk = p.evaluate_G(r.Xak, r.g[0], f)
        return 0

    # DPGOHash.cpp:446-581
    def mm_pgo(self):
        r, p, o = self.results, self.problem, self.options
        n0 = p.n[0]
        g, Df, f = r.g[0], r.Dfobj[0], r.f
        refined = (_div(r.gradFnorm * r.gradFnorm, r.fobj[0]) > o.accepted_delta) \
            and o.max_iterations > 0 and o.max_iterations_accepted > 0
        r.refined = refined
        r.Xakh = p.proximal(r.Xk, Df)
        R = r.Xakh[n0:]
        t = p.recover_translations(R, g)
        r.Xakh = np.vstack([t, R])
        if refined:
            res = self._tnt(r.Xakh, g)
            r.Xak = res["x"]
            r.Gk = res["f"]
        else:
            r.Xak = r.Xakh.copy()
            r.Gk = p.evaluate_G(r.Xak, r.g[0], f)
        return 0

    # DPGOHash.cpp:583-628
    def iterate(self):
        r, p = self.results, self.problem
        assert r.updated
        if self.options.scheme == SCHEME_AMM:
            self.amm_pgo()
        else:
            self.mm_pgo()
        r.iters += 1
        r.Xk[:(p.d + 1) * p.n[0]] = r.Xak
        r.updated = False
        return 0
