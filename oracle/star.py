"""Oracle: global objective / gradient, chordal initialisation and the
dist_pgo driver loop.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  * DPGOStar::evaluate_f / evaluate_grad    C++/DPGO/src/DPGOStar.cpp:713-829
  * DPGO::communicate                       C++/DPGO/include/DPGO/DPGO_utils.h:397-453
  * centralised chordal initialisation      C++/SESync/src/SESync_utils.cpp:573-652
    (SPQR least squares -> scipy sparse normal equations; same minimiser)
  * the driver loop                         C++/examples/dist_pgo.cpp:446-531
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .assemble import assemble_global
from .g2o import read_g2o_file, partition_measurements
from .hash import DPGOHash, Options
from .problem import (LOSS_GM, LOSS_HUBER, LOSS_NONE, LOSS_WELSCH,
                      project_to_SOdn, tangent_proj)


def _div(a, b):
    """IEEE division as in C++ (inf / nan instead of ZeroDivisionError): a node whose objective is exactly
    zero is still handled the way the reference handles it."""
    if b == 0.0:
        return float("nan") if a == 0.0 or a != a else math.copysign(float("inf"), a)
    return a / b


class GlobalProblem:
    """The evaluation half of DPGOStar (ctor DPGOStar.cpp:7-40)."""

    def __init__(self, num_poses, mm, num_nodes, options):
        self.options = options
        self.num_poses = num_poses
        self.d = mm.d
        _, self.measurements, self.g_index = partition_measurements(num_poses, mm, num_nodes)
        from .g2o import partition_index
        node_of, _ = partition_index(num_poses, num_nodes)
        intra_mask = node_of[mm.ipose] == node_of[mm.jpose]
        ia = np.nonzero(intra_mask)[0]
        ie = np.nonzero(~intra_mask)[0]
        self.intra, self.inter = mm.take(ia), mm.take(ie)
        self.M, self.B0, self.B1 = assemble_global(
            num_poses, self.d, self.intra, self.inter,
            mm.ipose[ia], mm.jpose[ia], mm.ipose[ie], mm.jpose[ie])

    def _rho_w(self, en2):
        o = self.options
        dl = o.loss_reg
        if o.loss == LOSS_HUBER:
            rs = np.sqrt(np.maximum(en2, dl))
            return np.minimum(2 * np.sqrt(dl) * rs - dl, en2), np.sqrt(dl) / rs
        if o.loss == LOSS_GM:
            return dl * en2 / (en2 + dl), dl * dl / (en2 + dl) ** 2
        if o.loss == LOSS_WELSCH:
            w = np.exp(-en2 / dl)
            return dl - dl * w, w
        raise ValueError("invalid loss")

    def evaluate_f(self, X):
        """DPGOStar.cpp:713-761."""
        if self.options.loss == LOSS_NONE:
            return 0.5 * float(np.sum(X * (self.M @ X)))
        d = self.d
        f = 0.5 * float(np.sum((self.B0 @ X) ** 2))
        m1 = len(self.inter)
        if m1:
            Err = self.B1 @ X
            en2 = np.sum(Err.reshape(m1, (d + 1) * d) ** 2, axis=1)
            rho, _ = self._rho_w(en2)
            f += 0.5 * float(np.sum(rho))
        return f

    def evaluate_grad(self, X):
        """DPGOStar.cpp:763-829."""
        d, n = self.d, self.num_poses
        if self.options.loss == LOSS_NONE:
            Df = self.M @ X
        else:
            Df = self.B0.T @ (self.B0 @ X)
            m1 = len(self.inter)
            if m1:
                Err = self.B1 @ X
                en2 = np.sum(Err.reshape(m1, (d + 1) * d) ** 2, axis=1)
                _, w = self._rho_w(en2)
                Df = Df + self.B1.T @ (np.repeat(w, d + 1)[:, None] * Err)
        grad = Df.copy()
        grad[n:] = tangent_proj(X[n:], Df[n:], d)
        return grad


def chordal_initialization(num_poses, mm):
    """SESync_utils.cpp:573-652: rotations minimise sum kappa |R_j - R_i R_ij|^2
    with R_0 = I (then per-block projection), translations minimise
    sum tau |t_j - t_i - R_i t_ij|^2 with t_0 = 0.  Returns X in the
    reference layout [t; R^T blocks]."""
    d, n, M = mm.d, num_poses, len(mm)
    I, J = mm.ipose, mm.jpose
    # unknown Y_i = R_i^T (d x d); residual sqrt(kappa) (R_ij^T Y_i - Y_j)
    rows, cols, vals = [], [], []
    e = np.arange(M)
    sk = np.sqrt(mm.kappa)
    for r in range(d):
        for c in range(d):
            rows.append(e * d + r)
            cols.append(I * d + c)
            vals.append(sk * mm.R[:, c, r])
        rows.append(e * d + r)
        cols.append(J * d + r)
        vals.append(-sk)
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(d * M, d * n)).tocsc()
    A0, Ar = A[:, :d], A[:, d:]
    rhs = -(A0 @ np.eye(d))
    Yr = spla.splu((Ar.T @ Ar).tocsc()).solve((Ar.T @ rhs))
    Y = np.vstack([np.eye(d), Yr])
    Y[d:] = project_to_SOdn(Y[d:], d)
    # translations: residual sqrt(tau) (x_i - x_j + t^T Y_i)
    st = np.sqrt(mm.tau)
    B = sp.coo_matrix((np.concatenate([st, -st]), (np.concatenate([e, e]), np.concatenate([I, J]))),
                      shape=(M, n)).tocsc()
    c = st[:, None] * np.einsum("ek,ekc->ec", mm.t, Y.reshape(n, d, d)[I])
    Br = B[:, 1:]
    xr = spla.splu((Br.T @ Br).tocsc()).solve(-(Br.T @ c))
    x = np.vstack([np.zeros((1, d)), xr])
    return np.vstack([x, Y])


class DistPGO:
    """The dist_pgo driver (C++/examples/dist_pgo.cpp:92-126, 446-531) with the
    centralised initialisation branch (:416-444)."""

    def __init__(self, filename, num_nodes, options=None, X0=None, mm=None, num_poses=None):
        self.options = options or Options.driver()
        if mm is None:
            num_poses, mm = read_g2o_file(filename)
        self.num_poses, self.mm, self.num_nodes = num_poses, mm, num_nodes
        self.d = d = mm.d
        _, self.measurements, self.g_index = partition_measurements(num_poses, mm, num_nodes)
        self.nodes = [DPGOHash(a, self.measurements[a], self.options) for a in range(num_nodes)]
        self.star = GlobalProblem(num_poses, mm, num_nodes, self.options)
        if X0 is None:
            X0 = chordal_initialization(num_poses, mm)
        self.X0 = X0
        self.offset = [min(gi.values()) if gi else 0 for gi in self.g_index]
        # split X0 per node (:435-443) and fill neighbour rows (DPGO::communicate, :446)
        Xs = []
        for a, nd in enumerate(self.nodes):
            n0 = nd.problem.n[0]
            o = self.g_index[a][0]
            Xs.append(np.vstack([X0[o:o + n0], X0[num_poses + d * o: num_poses + d * (o + n0)]]))
        for a, nd in enumerate(self.nodes):
            p = nd.problem
            n0, n1 = p.n
            Z = np.zeros(((d + 1) * (n0 + n1), d))
            Z[:(d + 1) * n0] = Xs[a]
            for beta, poses in p.info.index.items():
                if beta == a:
                    continue
                nb0 = self.nodes[beta].problem.n[0]
                for j, (_, k) in poses.items():
                    Z[(d + 1) * n0 + k] = Xs[beta][j]
                    r0 = (d + 1) * n0 + n1 + k * d
                    Z[r0:r0 + d] = Xs[beta][nb0 + j * d: nb0 + j * d + d]
            nd.initialize(Z)
            nd.update()
        self.trace = [self.evaluate()]

    def gather(self):
        """dist_pgo.cpp:502-511."""
        d, N = self.d, self.num_poses
        X = np.zeros(((d + 1) * N, d))
        for a, nd in enumerate(self.nodes):
            n0 = nd.problem.n[0]
            o = self.g_index[a][0]
            Xk = nd.results.Xk
            X[o:o + n0] = Xk[:n0]
            X[N + d * o: N + d * (o + n0)] = Xk[n0:n0 + d * n0]
        return X

    def evaluate(self):
        """(2F, 2|grad F|) as printed by the driver (dist_pgo.cpp:477-481)."""
        X = self.gather()
        return 2 * self.star.evaluate_f(X), 2 * float(np.linalg.norm(self.star.evaluate_grad(X)))

    def step(self, evaluate=True):
        for nd in self.nodes:
            nd.iterate()
        for nd in self.nodes:
            nd.communicate(self.nodes)
        for nd in self.nodes:
            nd.update()
        if evaluate:
            self.trace.append(self.evaluate())

    def run(self, iters, evaluate=True):
        for _ in range(iters):
            self.step(evaluate)
        return self.trace


class DPGOStar:
    """AMM-PGO* (master node aggregates the global objective).

    Restates DPGOStar::{initialize, iterate, communicate, update} and the per-node
    helpers {initialize_n, update_n, amm_pgo_n, mm_pgo_n, pm_pgo_n, communicate_n}
    (C++/DPGO/src/DPGOStar.cpp:107-711).  Intended loop (asserts at :318, :393):
    initialize(X); repeat { update(); iterate(); communicate() }."""

    def __init__(self, filename, num_nodes, options=None, mm=None, num_poses=None):
        from .hash import DPGOHash, Results
        self.options = options or Options.driver()
        if mm is None:
            num_poses, mm = read_g2o_file(filename)
        self.num_poses, self.mm, self.num_nodes, self.d = num_poses, mm, num_nodes, mm.d
        _, self.measurements, self.g_index = partition_measurements(num_poses, mm, num_nodes)
        # DPGOHash objects are used as containers of (problem, results); their own
        # update()/iterate() are not called
        self.nodes = [DPGOHash(a, self.measurements[a], self.options) for a in range(num_nodes)]
        self.star = GlobalProblem(num_poses, mm, num_nodes, self.options)
        self.Xk = self.Xkh = self.Xkp = None
        self.F = self.fobj = 0.0

    def _gid(self, node, pose):
        return self.g_index[node][pose]

    def _fill(self, a, X, own=True):
        """initialize_n / communicate_n (:215-313): rows of node a's Z from the global X."""
        nd = self.nodes[a]
        p, d, N = nd.problem, self.d, self.num_poses
        n0, n1 = p.n
        Z = nd.results.Xk
        for beta, poses in p.info.index.items():
            if beta == a and not own:
                continue
            for j, (blk, k) in poses.items():
                g = self._gid(beta, j)
                s = 0 if blk == 0 else (d + 1) * n0
                nn = n0 if blk == 0 else n1
                Z[s + k] = X[g]
                Z[s + nn + d * k: s + nn + d * k + d] = X[N + d * g: N + d * g + d]

    def _put(self, a, Xglob, Xa):
        """Write node a's own poses into a global matrix (:545-550)."""
        n0, d, N = self.nodes[a].problem.n[0], self.d, self.num_poses
        o = self.g_index[a][0]
        Xglob[o:o + n0] = Xa[:n0]
        Xglob[N + d * o: N + d * (o + n0)] = Xa[n0:]

    def initialize(self, X):
        from .hash import Results
        d = self.d
        for a, nd in enumerate(self.nodes):
            n0, n1 = nd.problem.n
            nd.results = Results()
            nd.results.Xk = np.zeros(((d + 1) * (n0 + n1), d))
            self._fill(a, X, own=True)
            nd.results.Xak = nd.results.Xk[:(d + 1) * n0].copy()
            nd.results.gamma = 0.0
            nd.results.updated = False
        self.Xk = X.copy()
        self.Xkh = np.zeros_like(X)
        self.Xkp = np.zeros_like(X)
        self.fobj = self.star.evaluate_f(self.Xk)
        self.F = self.fobj
        return 0

    def communicate(self):
        for a, nd in enumerate(self.nodes):
            self._fill(a, self.Xk, own=False)
            nd.results.updated = False
        return 0

    def update(self):
        """update_n for every node (:315-390): re-linearise from scratch every iteration."""
        import math
        for nd in self.nodes:
            r, p, o = nd.results, nd.problem, self.options
            if r.updated:
                continue
            it = r.iters
            r.X = [r.Xk.copy(), r.X[0]]
            if p.trivial:
                g, f = p.evaluate_none_g_and_f0(r.X[0])
                fobj = p.evaluate_G(r.Xak, g, f)
                Dfobj, gradF = p.full_Riemannian_gradient_G(r.Xak, g)
            else:
                if p.dynamic:      # Rescale::Dynamic (DPGOStar.cpp:350-355)
                    g, f, Dfobj, fobj, r.DfobjE, r.fobjE, r.rescale_count = p.evaluate_g_and_f0_rescale(
                        r.X[0], getattr(r, "rescale_count", 0), o.max_rescale_count)
                else:
                    g, f, Dfobj, fobj, r.DfobjE, r.fobjE = p.evaluate_g_and_f0(r.X[0])
                gradF = p.full_tangent_space_projection(r.Xak, Dfobj)
            r.Gk = fobj
            r.g = [g, r.g[0]]
            r.Dfobj = [Dfobj, r.Dfobj[0]]
            r.f = f
            r.fobj = [fobj, r.fobj[0]]
            r.gradFnorm = float(np.linalg.norm(gradF))
            if o.scheme == 1:
                r.s = [1.0, 1.0] if it == 0 else [r.s[1], 0.0]
                r.s[1] = 0.5 + 0.5 * math.sqrt(4.0 * r.s[0] * r.s[0] + 1.0)
                r.gamma = (r.s[0] - 1) / r.s[1]
            r.Fk = [fobj, fobj]
            r.updated = True
        return 0

    def _amm_n(self, a):
        nd = self.nodes[a]
        r, p, o = nd.results, nd.problem, self.options
        n0 = p.n[0]
        if r.iters == 0:
            Y, g, Df = r.Xk, r.g[0], r.Dfobj[0]
        else:
            Y = r.X[0] + r.gamma * (r.X[0] - r.X[1])
            if p.trivial:
                g = r.g[0] + r.gamma * (r.g[0] - r.g[1])
                Df = r.Dfobj[0] + r.gamma * (r.Dfobj[0] - r.Dfobj[1])
            else:
                g, Df = p.evaluate_g_and_Df(Y)
        r.refined = _div(r.gradFnorm * r.gradFnorm, r.fobj[0]) > o.accepted_delta       # :515-516
        r.Xakh = p.proximal(Y, Df)
        R = r.Xakh[n0:].copy()
        r.Xak = np.vstack([p.recover_translations(R, g), R])
        if r.refined:
            r.Xak = nd._tnt(r.Xak, g)["x"]
        self._put(a, self.Xkh, r.Xakh)
        self._put(a, self.Xkp, r.Xak)

    def _pm_n(self, a):
        nd = self.nodes[a]
        r = nd.results
        r.Xakh = nd.problem.proximal(r.Xk, r.Dfobj[0])
        self._put(a, self.Xkh, r.Xakh)

    def _mm_n(self, a):
        nd = self.nodes[a]
        r, p, o = nd.results, nd.problem, self.options
        n0 = p.n[0]
        g = r.g[0]
        refined = _div(r.gradFnorm * r.gradFnorm, r.fobj[0]) > o.accepted_delta
        R = r.Xakh[n0:].copy()
        r.Xak = np.vstack([p.recover_translations(R, g), R])
        if refined:
            res = nd._tnt(r.Xak, g)
            r.Xak, r.Gk = res["x"], res["f"]
        else:
            r.Gk = p.evaluate_G(r.Xak, g, r.f)
        self._put(a, self.Xkp, r.Xak)

    def iterate(self):
        """DPGOStar::iterate (:126-213)."""
        o = self.options
        for a in range(self.num_nodes):
            self._amm_n(a)
        fobjh = self.star.evaluate_f(self.Xkh)
        self.branches = []
        if fobjh > self.F - o.psi * float(np.sum((self.Xkh - self.Xk) ** 2)):
            self.branches.append("pm")
            for a in range(self.num_nodes):
                self._pm_n(a)
            fobjh = self.star.evaluate_f(self.Xkh)
        fobj = self.star.evaluate_f(self.Xkp)
        if fobj > self.F - o.psi * float(np.sum((self.Xkp - self.Xk) ** 2)):
            self.branches.append("mm")
            for a, nd in enumerate(self.nodes):
                self._mm_n(a)
                nd.results.s[1] = max(0.5 * nd.results.s[1], 1.0)
            fobj = self.star.evaluate_f(self.Xkp)
        if self.F - fobj < o.phi * (self.F - fobjh):
            self.branches.append("phi")
            for a, nd in enumerate(self.nodes):
                r, p = nd.results, nd.problem
                n0 = p.n[0]
                R = r.Xakh[n0:].copy()
                r.Xak = np.vstack([p.recover_translations(R, r.g[0]), R])
                self._put(a, self.Xkp, r.Xak)
            fobj = self.star.evaluate_f(self.Xkp)
        for nd in self.nodes:
            r, p = nd.results, nd.problem
            r.iters += 1
            r.Xk[:(self.d + 1) * p.n[0]] = r.Xak
            r.updated = False
        self.Xk, self.Xkp = self.Xkp, self.Xk
        self.fobj, self.fobjh = fobj, fobjh
        self.F = self.F * (1 - o.eta[0]) + fobj * o.eta[0]
        return 0

    def step(self):
        self.update()
        self.iterate()
        self.communicate()
