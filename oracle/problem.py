"""Oracle: per-node problem (surrogate operators) and SO(d)^n geometry.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates C++/DPGO/src/DPGOProblem.cpp, C++/DPGO/include/DPGO/DPGOProblem.h
and C++/DPGO/include/DPGO/SOdProduct.h on explicit scipy matrices.  X/Z use
the reference layout: rows [0,n) translations, rows [n + d*i, n + d*i + d)
the block R_i^T (DPGOProblem.h:167-171).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .assemble import assemble_node
from .g2o import generate_data_info

LOSS_NONE, LOSS_HUBER, LOSS_GM, LOSS_WELSCH = 0, 1, 2, 3
LOSS_NAMES = {"trivial": LOSS_NONE, "none": LOSS_NONE, "huber": LOSS_HUBER,
              "gm": LOSS_GM, "welsch": LOSS_WELSCH}


def project_to_SOdn(A, d):
    """Nearest rotation of every d x d row block of A (dn x d).

    Mathematical definition of project_to_SO{2,3,d}n
    (C++/DPGO/include/DPGO/DPGO_utils.h:485-565): U diag(1,..,det(UV^T)) V^T.
    For d = 2 the reference uses the closed form of
    C++/DPGO/include/DPGO/internal/project_to_SO2.h:3-18 with the guard
    c^2+s^2 >= 1e-32 (traits.cpp:10); it is the same projection."""
    n = A.shape[0] // d
    B = A.reshape(n, d, d)
    if d == 2:
        c = B[:, 0, 0] + B[:, 1, 1]
        s = B[:, 1, 0] - B[:, 0, 1]
        nrm2 = c * c + s * s
        ok = nrm2 >= 1e-32
        c = np.where(ok, c, 1.0)
        s = np.where(ok, s, 0.0)
        inv = 1.0 / np.sqrt(np.where(ok, nrm2, 1.0))
        c, s = c * inv, s * inv
        out = np.empty_like(B)
        out[:, 0, 0], out[:, 0, 1], out[:, 1, 0], out[:, 1, 1] = c, -s, s, c
        return out.reshape(n * d, d)
    U, _, Vt = np.linalg.svd(B)
    det = np.linalg.det(U) * np.linalg.det(Vt)
    U[:, :, -1] *= np.where(det > 0, 1.0, -1.0)[:, None]
    return (U @ Vt).reshape(n * d, d)


def sym_block_diag_product(A, B, C, d):
    """SOdProduct::SymBlockDiagProduct (SOdProduct.h:64-89):
    P_i = sym(C_i B_i^T) A_i."""
    n = A.shape[0] // d
    A3, B3, C3 = (M.reshape(n, d, d) for M in (A, B, C))
    Gm = C3 @ B3.transpose(0, 2, 1)
    S = 0.5 * (Gm + Gm.transpose(0, 2, 1))
    return (S @ A3).reshape(n * d, d)


def tangent_proj(Y, V, d):
    """SOdProduct::Proj (SOdProduct.h:96-103)."""
    return V - sym_block_diag_product(Y, Y, V, d)


class SpdSolver:
    """Exact SPD solve (stands in for Eigen::CholmodDecomposition,
    DPGO_types.h:27; call sites DPGOProblem.cpp:93,119)."""

    def __init__(self, A):
        self.lu = spla.splu(sp.csc_matrix(A), diag_pivot_thresh=0.0,
                            permc_spec="MMD_AT_PLUS_A",
                            options=dict(SymmetricMode=True))

    def solve(self, B):
        return self.lu.solve(np.ascontiguousarray(B))


class DPGOProblem:
    """DPGOProblem.cpp:11-125 (constructor) and the evaluate_* family."""

    MAX_RESCALE, MIN_RESCALE = 1.0, 0.01          # DPGOProblem.h:17-18

    def __init__(self, node, measurements, reg=1e-5, loss=LOSS_NONE,
                 reg_chol_precon_max_cond=1e6, loss_reg=1.0, preconditioner=True, dynamic=False):
        self.node = node
        self.info = generate_data_info(node, measurements)
        self.d = measurements.d
        self.n = self.info.n
        self.s = self.info.s
        self.m = self.info.m
        self.loss = loss
        self.loss_reg = loss_reg
        self.trivial = (loss == LOSS_NONE)          # "loss_ == None && SIMPLE"
        self.reg = reg
        # Rescale::Dynamic (DPGOProblem.cpp:46-84): the robust surrogate with one scale per inter-node edge,
        # all ones at construction
        self.dynamic = dynamic and not self.trivial
        self.scale = np.ones(self.m[1]) if self.dynamic else None
        self.mat = assemble_node(self.info, self.d, reg, self.trivial, self.scale)
        self.size0 = (self.d + 1) * self.n[0]
        self.L = SpdSolver(self.mat.Gtt)            # :93
        self.precon = None
        self.jacobi = None
        # (True: RegularizedCholesky, the value older callers pass; note True == 1 in Python)
        if preconditioner is not True and preconditioner == 1:
            self.jacobi = 1.0 / self.mat.GRR.diagonal()      # Preconditioner::Jacobi, :96-98
        elif preconditioner is True or preconditioner == 3:
            # RegularizedCholesky, :101-124 (Spectra, tol 1e-4, ncv 3)
            GRR = self.mat.GRR
            if GRR.shape[0] > 3:
                lam = spla.eigsh(GRR, k=1, which="LM", tol=1e-4, ncv=min(GRR.shape[0] - 1, 20),
                                 return_eigenvectors=False)[0]
            else:
                lam = np.linalg.eigvalsh(GRR.toarray())[-1]
            self.lambda_max = lam
            self.precon = SpdSolver(GRR + (lam / reg_chol_precon_max_cond) * sp.eye(GRR.shape[0]))

    # --- geometry -------------------------------------------------------
    def project(self, M):
        return project_to_SOdn(M, self.d)

    def recover_translations(self, R, g):
        """DPGOProblem.h:275-294: t = -G_tt^{-1}(g_t + G_tR R)."""
        n0 = self.n[0]
        return -self.L.solve(g[:n0] + self.mat.GtR @ R)

    def retract(self, Y, Ydot, g):
        """DPGOProblem.cpp:127-143."""
        n0 = self.n[0]
        Rp = self.project(Y[n0:] + Ydot)
        tp = self.recover_translations(Rp, g)
        return np.vstack([tp, Rp])

    def full_tangent_space_projection(self, Y, Ydot):
        """DPGOProblem.cpp:145-162."""
        n0 = self.n[0]
        out = Ydot.copy()
        out[n0:] = tangent_proj(Y[n0:], Ydot[n0:], self.d)
        return out

    def reduced_tangent_space_projection(self, Y, Ydot):
        """DPGOProblem.cpp:164-178."""
        return tangent_proj(Y[self.n[0]:], Ydot, self.d)

    # --- surrogate ------------------------------------------------------
    def evaluate_G(self, Y, g, f):
        """DPGOProblem.cpp:180-205: tr(Y^T (g + 1/2 G Y)) + f."""
        temp = g + 0.5 * (self.mat.G @ Y)
        return float(np.sum(Y * temp)) + f

    def _weights(self, err_norm2):
        """Loss re-weighting of DPGOProblem.cpp:647-675 -> (w_e, rho_sum)."""
        dl = self.loss_reg
        if self.loss == LOSS_NONE:
            return np.ones_like(err_norm2), float(np.sum(err_norm2))
        if self.loss == LOSS_HUBER:
            rescale = np.sqrt(np.maximum(err_norm2, dl))
            w = np.sqrt(dl) / rescale
            rho = np.minimum(2 * np.sqrt(dl) * rescale - dl, err_norm2)
            return w, float(np.sum(rho))
        if self.loss == LOSS_GM:
            w = dl * dl / (err_norm2 + dl) ** 2
            return w, float(dl * np.sum(err_norm2 / (err_norm2 + dl)))
        if self.loss == LOSS_WELSCH:
            w = np.exp(-err_norm2 / dl)
            return w, float(dl * len(err_norm2) - dl * np.sum(w))
        raise ValueError("Invalid loss kernel")

    def evaluate_E(self, Z):
        """DPGOProblem.cpp:634-681: (DfobjE, fobjE, weights)."""
        d, m1 = self.d, self.m[1]
        Err = self.mat.B1 @ Z                                   # ((d+1) m1, d)
        en2 = np.sum(Err.reshape(m1, (d + 1) * d) ** 2, axis=1) if m1 else np.zeros(0)
        w, rho_sum = self._weights(en2)
        fobjE = 0.5 * rho_sum
        W = np.repeat(w, d + 1)[:, None]
        DfobjE = self.mat.B1.T @ (W * Err)
        return DfobjE, fobjE, w

    def evaluate_g(self, Z):
        """DPGOProblem.cpp:683-725."""
        if self.trivial:
            return self.mat.S @ Z
        DfobjE, _, _ = self.evaluate_E(Z)
        return DfobjE[:self.size0] - self.mat.D @ Z[:self.size0]

    def evaluate_Df(self, Z, g):
        """DPGOProblem.cpp:727-749."""
        return g + self.mat.G @ Z[:self.size0]

    def evaluate_g_and_Df(self, Z):
        g = self.evaluate_g(Z)
        return g, self.evaluate_Df(Z, g)

    def evaluate_none_g_and_f0(self, Z):
        """DPGOProblem.cpp:269-287."""
        g = self.mat.S @ Z
        f0 = 0.5 * float(np.sum(Z * (self.mat.P0 @ Z)))
        return g, f0

    def evaluate_none_g_and_f(self, Z, Z0, G):
        """DPGOProblem.cpp:516-542."""
        g = self.mat.S @ Z
        Y = Z - Z0
        fobj = G + 0.5 * float(np.sum(Y * (self.mat.Q @ Y)))
        f = fobj + 0.5 * float(np.sum(Z * (self.mat.P @ Z)))
        return g, f, fobj

    def evaluate_g_and_f0(self, Z):
        """DPGOProblem.cpp:222-267 -> g, f0, Dfobj, fobj, DfobjE, fobjE."""
        X = Z[:self.size0]
        DfobjE, fobjE, _ = self.evaluate_E(Z)
        g = DfobjE[:self.size0].copy()
        temp = self.mat.D @ X
        g -= temp
        temp = 0.5 * temp - DfobjE[:self.size0]
        f0 = 0.5 * fobjE + float(np.sum(X * temp))
        temp = self.mat.G @ X
        Dfobj = g + temp
        fobj = f0 + float(np.sum(X * (0.5 * temp + g)))
        return g, f0, Dfobj, fobj, DfobjE, fobjE

    def update_quadratic_mat(self, scale):
        """DPGOProblem.cpp:751-840 + L_.factorize (:315, :479).  The preconditioner and lambda_max keep the
        values of the constructor (the reference never recomputes them)."""
        self.scale = np.asarray(scale, np.float64).copy()
        self.mat = assemble_node(self.info, self.d, self.reg, self.trivial, self.scale)
        self.L = SpdSolver(self.mat.Gtt)

    def _maybe_rescale(self, w, rescale_count, max_rescale_count):
        """The test of DPGOProblem.cpp:300-321 / :464-485 -> new rescale_count."""
        if (rescale_count >= max_rescale_count) or bool(np.any(w > self.scale)):
            self.update_quadratic_mat(np.clip(1.25 * w, self.MIN_RESCALE, self.MAX_RESCALE))
            return 0
        return rescale_count + 1

    def evaluate_g_and_f0_rescale(self, Z, rescale_count, max_rescale_count):
        """DPGOProblem.cpp:289-358 -> g, f0, Dfobj, fobj, DfobjE, fobjE, rescale_count."""
        X = Z[:self.size0]
        DfobjE, fobjE, w = self.evaluate_E(Z)
        if self.dynamic:
            rescale_count = self._maybe_rescale(w, rescale_count, max_rescale_count)
        g = DfobjE[:self.size0].copy()
        temp = self.mat.D @ X
        g -= temp
        temp = 0.5 * temp - DfobjE[:self.size0]
        f0 = 0.5 * fobjE + float(np.sum(X * temp))
        temp = self.mat.G @ X
        Dfobj = g + temp
        fobj = f0 + float(np.sum(X * (0.5 * temp + g)))
        return g, f0, Dfobj, fobj, DfobjE, fobjE, rescale_count

    def evaluate_g_and_f_rescale(self, Z, Z0, G, DfobjE0, fobjE0, rescale_count, max_rescale_count):
        """DPGOProblem.cpp:426-514 (robust branch): the majorisation gap uses the OLD Q, then the surrogate may
        be rescaled, then g, Dfobj, f use the new D and G."""
        X = Z[:self.size0]
        Y = Z - Z0
        temp = DfobjE0 + 0.5 * (self.mat.Q @ Y)
        fobj = G - 0.5 * fobjE0 - 0.5 * float(np.sum(Y * temp))
        DfobjE, fobjE, w = self.evaluate_E(Z)
        fobj += 0.5 * fobjE
        if self.dynamic:
            rescale_count = self._maybe_rescale(w, rescale_count, max_rescale_count)
        g = DfobjE[:self.size0] - self.mat.D @ X
        temp = self.mat.G @ X
        Dfobj = g + temp
        f = fobj - float(np.sum(X * (0.5 * temp + g)))
        return g, f, Dfobj, fobj, DfobjE, fobjE, rescale_count

    def evaluate_g_and_f(self, Z, Z0, G, DfobjE0, fobjE0):
        """DPGOProblem.cpp:360-424 (robust branch) -> g, f, Dfobj, fobj, DfobjE, fobjE."""
        X = Z[:self.size0]
        Y = Z - Z0
        temp = DfobjE0 + 0.5 * (self.mat.Q @ Y)
        fobj = G - 0.5 * fobjE0 - 0.5 * float(np.sum(Y * temp))
        DfobjE, fobjE, _ = self.evaluate_E(Z)
        fobj += 0.5 * fobjE
        g = DfobjE[:self.size0] - self.mat.D @ X
        temp = self.mat.G @ X
        Dfobj = g + temp
        f = fobj - float(np.sum(X * (0.5 * temp + g)))
        return g, f, Dfobj, fobj, DfobjE, fobjE

    def full_Riemannian_gradient_G(self, Y, g):
        """DPGOProblem.h:356-376 -> (nablaF, gradF)."""
        nabla = g + self.mat.G @ Y
        return nabla, self.full_tangent_space_projection(Y, nabla)

    def reduced_Euclidean_gradient_G(self, Y, g):
        """DPGOProblem.h:380-393."""
        n0 = self.n[0]
        return g[n0:] + self.mat.G[n0:] @ Y

    def hessian_vector_product(self, Y, nablaF_Y, Ydot):
        """DPGOProblem.cpp:552-577."""
        n0 = self.n[0]
        R = Y[n0:]
        tdot = -self.L.solve(self.mat.GtR @ Ydot)
        E = self.mat.GRt @ tdot + self.mat.GRR @ Ydot
        E -= sym_block_diag_product(Ydot, R, nablaF_Y, self.d)
        return tangent_proj(R, E, self.d)

    def precondition(self, Y, Ydot):
        """DPGOProblem.cpp:579-598 (RegularizedCholesky)."""
        if self.jacobi is not None:                       # :583-585
            return self.reduced_tangent_space_projection(Y, self.jacobi[:, None] * Ydot)
        if self.precon is None:
            return Ydot
        return self.reduced_tangent_space_projection(Y, self.precon.solve(Ydot))

    def proximal(self, Z, Df):
        """DPGOProblem.cpp:600-632."""
        n0, d = self.n[0], self.d
        t0 = Z[:n0]
        R0 = Z[n0:n0 + d * n0]
        if self.trivial:
            M = self.mat.U @ Z
        else:
            M = -Df[n0:] + self.mat.N.T @ Df[:n0] + self.mat.V @ R0
        R = self.project(M)
        t = t0 - self.mat.N @ (R - R0) - self.mat.T[:, None] * Df[:n0]
        return np.vstack([t, R])
