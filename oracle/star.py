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
