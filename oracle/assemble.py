"""Oracle: explicit sparse surrogate matrices of one node.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the triplet assembly of
  * simplify_quadratic_data_matrix (trivial loss)  C++/DPGO/src/DPGO_utils.cpp:1398-2288
  * simplify_regular_data_matrix (Static rescale)   C++/DPGO/src/DPGO_utils.cpp:2290-2967
  * construct_data_matrix (global M, B0, B1)        C++/DPGO/src/DPGO_utils.cpp:440-718
with scipy COO matrices.  Every matrix is written in terms of the per-edge
"edge Hessian" E_e, the 2(d+1) x 2(d+1) matrix of the quadratic form
    tau*|x_i - x_j + t^T Y_i|^2 + kappa*|R^T Y_i - Y_j|_F^2
in the local ordering [t_i, R_i rows, t_j, R_j rows]; bd(E) keeps its two
diagonal blocks, od(E) the coupling blocks and E+ = bd(E) - od(E).  The
correspondence with the reference's triplet lists is spelled out next to each
matrix below.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .g2o import local_rows


def edge_hessians(meas):
    """E_e for every measurement: (M, 2(d+1), 2(d+1)).

    DPGO_utils.cpp:1541-1641 (identical in every builder, e.g. :500-550)."""
    M, d = len(meas), meas.d
    D1 = d + 1
    E = np.zeros((M, 2 * D1, 2 * D1))
    tau, kap, t, R = meas.tau, meas.kappa, meas.t, meas.R
    E[:, 0, 0] = tau
    E[:, 0, D1] = -tau
    E[:, D1, 0] = -tau
    E[:, D1, D1] = tau
    for k in range(d):
        E[:, 0, 1 + k] = tau * t[:, k]
        E[:, 1 + k, 0] = tau * t[:, k]
        E[:, D1, 1 + k] = -tau * t[:, k]
        E[:, 1 + k, D1] = -tau * t[:, k]
        E[:, 1 + k, 1 + k] += kap
        E[:, D1 + 1 + k, D1 + 1 + k] += kap
    for r in range(d):
        for c in range(d):
            E[:, 1 + r, 1 + c] += tau * t[:, r] * t[:, c]
            E[:, 1 + r, D1 + 1 + c] = -kap * R[:, r, c]
            E[:, D1 + 1 + r, 1 + c] = -kap * R[:, c, r]
    return E


def _edge_index(info, meas):
    """Global row index of every local slot of E_e: (M, 2(d+1)); own-mask (M, 2)."""
    d = meas.d
    ti, ri, tj, rj, bi, bj = local_rows(info, meas, d)
    M = len(meas)
    idx = np.empty((M, 2 * (d + 1)), np.int64)
    idx[:, 0] = ti
    idx[:, d + 1] = tj
    for k in range(d):
        idx[:, 1 + k] = ri + k
        idx[:, d + 2 + k] = rj + k
    own = np.stack([bi == 0, bj == 0], axis=1)
    return idx, own


class _Coo:
    def __init__(self, shape):
        self.shape = shape
        self.r, self.c, self.v = [], [], []

    def add(self, rows, cols, vals):
        self.r.append(np.asarray(rows).ravel())
        self.c.append(np.asarray(cols).ravel())
        self.v.append(np.asarray(vals, dtype=np.float64).ravel())

    def add_blocks(self, idx, E, rsel, csel, scale=1.0, emask=None):
        """Scatter E[:, rsel, csel] at (idx[:, rsel], idx[:, csel])."""
        if emask is not None:
            idx, E = idx[emask], E[emask]
        if len(idx) == 0:
            return
        rows = idx[:, rsel][:, :, None] + 0 * idx[:, csel][:, None, :]
        cols = 0 * idx[:, rsel][:, :, None] + idx[:, csel][:, None, :]
        self.add(rows, cols, scale * E[:, rsel][:, :, csel])

    def build(self):
        if not self.r:
            return sp.csr_matrix(self.shape)
        m = sp.coo_matrix((np.concatenate(self.v),
                           (np.concatenate(self.r), np.concatenate(self.c))),
                          shape=self.shape)
        return m.tocsr()


def _b_rows(meas, idx, ncols):
    """Residual matrix rows of DPGO_utils.cpp:1643-1677 / :2173-2207."""
    M, d = len(meas), meas.d
    D1 = d + 1
    B = _Coo((D1 * M, ncols))
    if M == 0:
        return B.build()
    l = D1 * np.arange(M)
    st, sk = np.sqrt(meas.tau), np.sqrt(meas.kappa)
    B.add(l, idx[:, 0], st)
    B.add(l, idx[:, D1], -st)
    for k in range(d):
        B.add(l, idx[:, 1 + k], st * meas.t[:, k])
        B.add(l + k + 1, idx[:, D1 + 1 + k], -sk)
    for r in range(d):
        for c in range(d):
            B.add(l + r + 1, idx[:, 1 + c], sk * meas.R[:, c, r])
    return B.build()


class NodeMatrices:
    pass


def assemble_node(info, d, xi, trivial, scale=None):
    """All matrices of one node.  trivial=True follows
    simplify_quadratic_data_matrix, trivial=False simplify_regular_data_matrix
    (Static rescale).  scale (one number per inter-node edge) selects the Dynamic
    variant (DPGO_utils.cpp:2969-3903 + DPGOProblem::update_quadratic_mat,
    DPGOProblem.cpp:751-840): every inter-edge contribution to G, D, Q and the proximal
    majoriser H (columns of E_ / F_, :3518-3585) is multiplied by its edge's scale, and the
    majoriser carries 0.5 xi instead of 1.5 xi (:3621, :3634 against :2222-2241)."""
    n0, n1 = info.n
    D1 = d + 1
    N0, NZ = D1 * n0, D1 * (n0 + n1)
    A, I = slice(0, D1), slice(D1, 2 * D1)      # tail / head slots of E
    ALL = slice(0, 2 * D1)

    Ei = edge_hessians(info.intra)
    ii, _ = _edge_index(info, info.intra)
    Ee = edge_hessians(info.inter)
    dynamic = scale is not None
    if dynamic:
        Ee = Ee * np.asarray(scale, np.float64)[:, None, None]
    ie, own = _edge_index(info, info.inter)
    tail_own, head_own = own[:, 0], own[:, 1]

    def bd(E):
        out = np.zeros_like(E)
        out[:, A, A] = E[:, A, A]
        out[:, I, I] = E[:, I, I]
        return out

    Eip = 2 * bd(Ei) - Ei                       # E+ = bd - od
    Eep = 2 * bd(Ee) - Ee

    G = _Coo((N0, N0))
    D = _Coo((N0, N0))
    H = _Coo((N0, N0))
    Q = _Coo((NZ, NZ))
    # intra part of G (:1541-1641), H = 2*bd(E) (:1679-1738)
    G.add_blocks(ii, Ei, ALL, ALL)
    H.add_blocks(ii, Ei, A, A, 2.0)
    H.add_blocks(ii, Ei, I, I, 2.0)
    # inter part: 2*bd(E) on the own endpoint (:1964-2028, :2097-2127)
    for Mx in (G, D, H):
        Mx.add_blocks(ie, Ee, A, A, 2.0, tail_own)
        Mx.add_blocks(ie, Ee, I, I, 2.0, head_own)
    own_diag = np.arange(N0)
    G.add(own_diag, own_diag, np.full(N0, xi))            # :2212-2233
    D.add(own_diag, own_diag, np.full(N0, xi))
    H.add(own_diag, own_diag, np.full(N0, (0.5 if dynamic else 1.5) * xi))

    out = NodeMatrices()
    out.trivial = trivial
    if trivial:
        S = _Coo((N0, NZ))
        P = _Coo((NZ, NZ))
        P0 = _Coo((NZ, NZ))
        V = _Coo((N0, NZ))
        # Q = -1/2 E+ per inter edge (:1864-1962), -xi on own diagonal (:2217,2231)
        Q.add_blocks(ie, Eep, ALL, ALL, -0.5)
        Q.add(own_diag, own_diag, np.full(N0, -xi))
        # P0 = +1/2 E+ (:1872-1961), +xi (:2219,2233)
        P0.add_blocks(ie, Eep, ALL, ALL, 0.5)
        P0.add(own_diag, own_diag, np.full(N0, xi))
        # P = -E (intra, :1552-1640) - od(E) (inter, :1869-1948) + xi
        P.add_blocks(ii, Ei, ALL, ALL, -1.0)
        P.add_blocks(ie, Ee, A, I, -1.0)
        P.add_blocks(ie, Ee, I, A, -1.0)
        P.add(own_diag, own_diag, np.full(N0, xi))
        # S = -E+[own endpoint rows, :] (:1971-2035, :2105-2134) - xi
        S.add_blocks(ie, Eep, A, ALL, -1.0, tail_own)
        S.add_blocks(ie, Eep, I, ALL, -1.0, head_own)
        S.add(own_diag, own_diag, np.full(N0, -xi))
        # V = -E+ (intra :1687-1753; inter own rows :2044-2094, :2143-2168) - 1.5 xi
        V.add_blocks(ii, Eip, ALL, ALL, -1.0)
        V.add_blocks(ie, Eep, A, ALL, -1.0, tail_own)
        V.add_blocks(ie, Eep, I, ALL, -1.0, head_own)
        V.add(own_diag, own_diag, np.full(N0, -1.5 * xi))
        out.S, out.P, out.P0, out.Vfull = S.build(), P.build(), P0.build(), V.build()
    else:
        # robust Q = 2*bd(E) on BOTH endpoints (:2711-2746) + 2 xi own (:2910-2921)
        Q.add_blocks(ie, Ee, A, A, 2.0)
        Q.add_blocks(ie, Ee, I, I, 2.0)
        Q.add(own_diag, own_diag, np.full(N0, 2.0 * xi))

    out.G, out.D, out.H, out.Q = G.build(), D.build(), H.build(), Q.build()
    out.Gtt = out.G[:n0, :n0].tocsr()
    out.GtR = out.G[:n0, n0:].tocsr()
    out.GRt = out.G[n0:, :n0].tocsr()
    out.GRR = out.G[n0:, n0:].tocsr()
    out.B0 = _b_rows(info.intra, ii, NZ)
    out.B1 = _b_rows(info.inter, ie, NZ)

    # T = diag(H_tt)^-1, N = T*H_tR  (:2280-2282 / :2958-2960)
    Tdiag = 1.0 / out.H[:n0, :n0].diagonal()
    out.T = Tdiag
    HtR = out.H[:n0, n0:].tocsr()
    out.N = sp.diags(Tdiag) @ HtR
    if trivial:
        # U = N^T V_t - V_R   (:2284-2285)
        out.U = (out.N.T @ out.Vfull[:n0]) - out.Vfull[n0:]
        out.U = out.U.tocsr()
    else:
        # V = H_RR - H_Rt T H_tR   (:2962-2964)
        out.V = (out.H[n0:, n0:] - out.H[n0:, :n0] @ (sp.diags(Tdiag) @ HtR)).tocsr()
    return out


def assemble_global(num_poses, d, intra, inter, gi_intra, gj_intra, gi_inter, gj_inter):
    """construct_data_matrix (DPGO_utils.cpp:440-718): M, B0, B1 on the
    global ordering X = [t_0..t_{N-1}; R_0^T; ...; R_{N-1}^T]."""
    D1 = d + 1
    NX = D1 * num_poses

    def gidx(gi, gj):
        M = len(gi)
        idx = np.empty((M, 2 * D1), np.int64)
        idx[:, 0] = gi
        idx[:, D1] = gj
        for k in range(d):
            idx[:, 1 + k] = num_poses + gi * d + k
            idx[:, D1 + 1 + k] = num_poses + gj * d + k
        return idx

    Mx = _Coo((NX, NX))
    ALL = slice(0, 2 * D1)
    ii = gidx(np.asarray(gi_intra, np.int64), np.asarray(gj_intra, np.int64))
    ie = gidx(np.asarray(gi_inter, np.int64), np.asarray(gj_inter, np.int64))
    if len(intra):
        Mx.add_blocks(ii, edge_hessians(intra), ALL, ALL)
    if len(inter):
        Mx.add_blocks(ie, edge_hessians(inter), ALL, ALL)
    B0 = _b_rows(intra, ii, NX) if len(intra) else sp.csr_matrix((0, NX))
    B1 = _b_rows(inter, ie, NX) if len(inter) else sp.csr_matrix((0, NX))
    return Mx.build(), B0, B1
