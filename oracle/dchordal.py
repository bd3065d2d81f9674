"""Oracle: distributed chordal initialisation (the `--dist_init true` branch of dist_pgo).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity unpinned: the reference's DChordal needs Eigen /
CHOLMOD / SE-Sync and ships no tests; this file restates its arithmetic on explicit numpy / scipy matrices.

Restates
  * the driver schedule                      C++/examples/dist_pgo.cpp:144-416
  * DChordal / DChordal_R / DChordal_t       C++/DChordal/src/DChordal.cpp:8-187,
                                             C++/DChordal/include/DChordal/DChordal.h:26-84,
                                             C++/DChordal/include/DChordal/DChordalProblem.h:128-246,
                                             C++/DChordal/src/DChordalProblem.cpp:51-104
  * DChordalReduced / _R / _t                C++/DChordal/src/DChordalReduced.cpp:38-183,
                                             C++/DChordal/include/DChordal/DChordalReduced.h:24-60,
                                             C++/DChordal/include/DChordal/DChordalReducedProblem.h:150-261,
                                             C++/DChordal/src/DChordalReducedProblem.cpp:42-115
  * the data matrices                        C++/DChordal/src/DChordal_utils.cpp:30-65 (n_index), :67-309 (reduced R),
                                             :311-363 (recover t), :365-603 (reduced t), :605-913 (R), :915-1204 (t)
  * communicate / n_communicate / evaluate_f C++/DChordal/include/DChordal/DChordal_utils.h:129-261

Stage 0 of the reference is a per-node SE-Sync solve (DChordal_utils.cpp:11-28), a third-party certifiable solver
that is out of scope.  Its stand-in here and in the product is the same deterministic rule:
chordal initialisation of the node's intra-node subgraph followed by LOCAL_ITERS iterations of MM-PGO with the
truncated-Newton refinement forced on (a Riemannian Newton-CG on the exact local objective: with no inter-node
edges the surrogate is the objective).  `local_solutions` lets a test inject stage-0 poses instead.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .g2o import Measurements, generate_data_info
from .problem import project_to_SOdn

LOCAL_ITERS = 30          # MM-PGO + TNT iterations of the stage-0 stand-in
REG_G = 1e-12             # DChordal::Options::reg_G (DChordal_types.h:50)
SCHEDULE = (100, 400, 150, 250)   # dist_pgo.cpp:205, 274, 344, 383


def project_to_SOd(M):
    """DChordal_utils.h:270-289 (one d x d block)."""
    U, _, Vt = np.linalg.svd(M)
    if np.linalg.det(U) * np.linalg.det(Vt) <= 0:
        U = U.copy()
        U[:, -1] *= -1
    return U @ Vt


def n_index_of(info, a):
    """DChordal_utils.cpp:47-60: own node 0, neighbour nodes 1.. in map (sorted) order."""
    out, count = {}, 1
    for b in sorted(info.index):
        if b == a:
            out[b] = 0
        else:
            out[b] = count
            count += 1
    return out


def _spd_solve(A):
    lu = spla.splu(sp.csc_matrix(A))
    return lambda b: lu.solve(np.asarray(b))


class _Nesterov:
    """The iterate / update pair shared by DChordal (DChordal.cpp:79-152) and DChordalReduced
    (DChordalReduced.cpp:114-183): X[k], s_k, Y = (1 + gamma) X[k] - gamma X[k-1], Xak = solve(g_ + S Y)."""

    def initialize(self, X):
        self.Xk = X.copy()
        self.Xak = X[:self.p * self.n_own].copy()
        self.X = [None]
        self.s = [1.0]
        self.iters = 0

    def update(self):
        it = self.iters
        while len(self.X) < it + 1:
            self.X.append(None)
        self.X[it] = self.Xk.copy()

    def iterate(self):
        it = self.iters
        s0 = self.s[it]
        s1 = 0.5 + 0.5 * math.sqrt(4.0 * s0 * s0 + 1.0)
        gamma = (s0 - 1) / s1
        self.s.append(s1)
        Y = self.X[0] if it == 0 else (1.0 + gamma) * self.X[it] - gamma * self.X[it - 1]
        self.Xak = self.solve(Y, self.evaluate_g(Y))
        self.Xk[:self.p * self.n_own] = self.Xak
        self.iters = it + 1

    def objective(self):
        """One node's term of evaluate_f (DChordal_utils.h:129-140)."""
        return float(np.sum((self.B @ self.Xk + self.b) ** 2))


def _edge_slots(info, a, e, d):
    """(block, local index) of tail and head of inter edge e, and which of them is the local pose."""
    m = info.inter
    bi, ki = info.index[int(m.inode[e])][int(m.ipose[e])]
    bj, kj = info.index[int(m.jnode[e])][int(m.jpose[e])]
    return (bi, ki), (bj, kj), int(m.inode[e]) == a


class ReducedR(_Nesterov):
    """DChordalReduced_R: one d x d block per node (DChordal_utils.cpp:67-309)."""

    def __init__(self, a, meas, xi=REG_G):
        self.a, self.info = a, generate_data_info(a, meas)
        self.d = meas.d
        self.p, self.n_own = self.d, 1
        self.n_index = n_index_of(self.info, a)
        self.nn = len(self.n_index) - 1
        self.xi = xi

    def setup(self, X):
        info, d, a = self.info, self.d, self.a
        n, s = info.n, info.s
        nn1, M = self.nn + 1, len(info.inter)
        G, S = np.zeros((d, d)), np.zeros((d, nn1 * d))
        B = np.zeros((M * d, nn1 * d))
        for e in range(M):
            (bi, ki), (bj, kj), tail_local = _edge_slots(info, a, e, d)
            Ri = X[(d + 1) * s[bi] + n[bi] + d * ki: (d + 1) * s[bi] + n[bi] + d * ki + d]
            Rj = X[(d + 1) * s[bj] + n[bj] + d * kj: (d + 1) * s[bj] + n[bj] + d * kj + d]
            nR = Ri.T @ info.inter.R[e] @ Rj                      # :255-258
            kap = info.inter.kappa[e]
            ni = (0, self.n_index[int(info.inter.jnode[e])]) if tail_local else (self.n_index[int(info.inter.inode[e])], 0)
            B[e * d:e * d + d, ni[0] * d: ni[0] * d + d] += math.sqrt(kap) * nR.T
            B[e * d:e * d + d, ni[1] * d: ni[1] * d + d] -= math.sqrt(kap) * np.eye(d)
            G += 2 * kap * np.eye(d)
            S[:, :d] -= kap * np.eye(d)
            if tail_local:
                S[:, ni[1] * d: ni[1] * d + d] -= kap * nR
            else:
                S[:, ni[0] * d: ni[0] * d + d] -= kap * nR.T
        G += self.xi * np.eye(d)
        S[:, :d] -= self.xi * np.eye(d)
        self.G, self.S, self.B, self.b = G, S, B, np.zeros((M * d, d))
        self.Ginv = np.linalg.inv(G)                              # DChordalReducedProblem.cpp:72

    def evaluate_g(self, Y):
        return self.S @ Y

    def solve(self, Y, g):
        return -self.Ginv @ g


class ReducedT(_Nesterov):
    """DChordalReduced_t: one translation per node (DChordal_utils.cpp:311-603)."""

    def __init__(self, a, meas, xi=REG_G):
        self.a, self.info = a, generate_data_info(a, meas)
        self.d = d = meas.d
        self.p, self.n_own = 1, 1
        self.n_index = n_index_of(self.info, a)
        self.nn = len(self.n_index) - 1
        self.xi = xi
        # precompute_data_matrix_recover_t (:311-363): pose ids are used as indices; pose 0 pinned by +100
        m, n0 = self.info.intra, self.info.n[0]
        I, J = m.ipose, m.jpose
        L = sp.coo_matrix((np.concatenate([m.tau, m.tau, -m.tau, -m.tau]),
                           (np.concatenate([I, J, I, J]), np.concatenate([I, J, J, I]))), shape=(n0, n0)).tolil()
        L[0, 0] += 100
        rows, cols, vals = [], [], []
        for k in range(d):
            rows += [I, J]
            cols += [I * d + k, I * d + k]
            vals += [m.tau * m.t[:, k], -m.tau * m.t[:, k]]
        self.P = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n0, n0 * d)).tocsr()
        self.LL = _spd_solve(L.tocsc())

    def recover_translations(self, R):
        """DChordalReducedProblem.h:251-261."""
        t = -self.LL(self.P @ R)
        return t - t[0]

    def setup(self, X, nR):
        info, d, a = self.info, self.d, self.a
        n, s = info.n, info.s
        nn1, M = self.nn + 1, len(info.inter)
        G, S, g = 0.0, np.zeros((1, nn1)), np.zeros((1, d))
        B, b = np.zeros((M, nn1)), np.zeros((M, d))
        for e in range(M):
            (bi, ki), (bj, kj), tail_local = _edge_slots(info, a, e, d)
            Ri = X[(d + 1) * s[bi] + n[bi] + d * ki: (d + 1) * s[bi] + n[bi] + d * ki + d]
            ti = X[(d + 1) * s[bi] + ki] + Ri.T @ info.inter.t[e]
            tj = X[(d + 1) * s[bj] + kj]
            tau = info.inter.tau[e]
            ni = (0, self.n_index[int(info.inter.jnode[e])]) if tail_local else (self.n_index[int(info.inter.inode[e])], 0)
            nt = nR[ni[0] * d: ni[0] * d + d].T @ ti - nR[ni[1] * d: ni[1] * d + d].T @ tj
            st = math.sqrt(tau)
            G += 2 * tau
            S[0, 0] -= tau
            if tail_local:
                B[e, ni[0]] += st
                B[e, ni[1]] -= st
                b[e] = st * nt
                S[0, ni[1]] -= tau
                g += tau * nt
            else:
                B[e, ni[0]] -= st
                B[e, ni[1]] += st
                b[e] = -st * nt
                S[0, ni[0]] -= tau
                g -= tau * nt
        G += self.xi
        S[0, 0] -= self.xi
        self.G, self.S, self.g_, self.B, self.b = G, S, g, B, b

    def evaluate_g(self, Y):
        return self.g_ + self.S @ Y

    def solve(self, Y, g):
        return -g / self.G


class ChordalR(_Nesterov):
    """DChordal_R (DChordal_utils.cpp:605-913, DChordalProblem.cpp:51-70, DChordalProblem.h:163-228)."""

    def __init__(self, a, meas, xi=REG_G):
        self.a, self.info = a, generate_data_info(a, meas)
        self.d = meas.d
        self.p, self.n_own = self.d, self.info.n[0]

    def setup(self):
        info, d, a = self.info, self.d, self.a
        n0, n1 = info.n
        rG, cG, vG, rS, cS, vS, rB, cB, vB = ([] for _ in range(9))

        def blk(rows, cols, vals, r0, c0, Mblk):
            for r in range(d):
                for c in range(d):
                    rows.append(r0 + r); cols.append(c0 + c); vals.append(Mblk[r, c])
        Id = np.eye(d)
        m = info.intra
        for e in range(len(m)):
            i, j = info.index[a][int(m.ipose[e])][1], info.index[a][int(m.jpose[e])][1]
            kap, R = m.kappa[e], m.R[e]
            blk(rG, cG, vG, i * d, i * d, kap * Id)
            blk(rG, cG, vG, j * d, j * d, kap * Id)
            blk(rG, cG, vG, i * d, j * d, -kap * R)
            blk(rG, cG, vG, j * d, i * d, -kap * R.T)
            blk(rB, cB, vB, e * d, i * d, math.sqrt(kap) * R.T)
            blk(rB, cB, vB, e * d, j * d, -math.sqrt(kap) * Id)
        m, M0 = info.inter, len(info.intra)
        for e in range(len(m)):
            (bi, ki), (bj, kj), tail_local = _edge_slots(info, a, e, d)
            ci, cj = d * info.s[bi] + ki * d, d * info.s[bj] + kj * d
            kap, R = m.kappa[e], m.R[e]
            if tail_local:
                blk(rG, cG, vG, ci, ci, 2 * kap * Id)
                blk(rS, cS, vS, ci, ci, -kap * Id)
                blk(rS, cS, vS, ci, cj, -kap * R)
            else:
                blk(rG, cG, vG, cj, cj, 2 * kap * Id)
                blk(rS, cS, vS, cj, cj, -kap * Id)
                blk(rS, cS, vS, cj, ci, -kap * R.T)
            blk(rB, cB, vB, (M0 + e) * d, ci, math.sqrt(kap) * R.T)
            blk(rB, cB, vB, (M0 + e) * d, cj, -math.sqrt(kap) * Id)
        # (+xi and -xi on the diagonal of G cancel, :891-894)
        G = sp.coo_matrix((vG, (rG, cG)), shape=(n0 * d, n0 * d)).tocsc()
        self.S = sp.coo_matrix((vS, (rS, cS)), shape=(n0 * d, (n0 + n1) * d)).tocsr()
        self.B = sp.coo_matrix((vB, (rB, cB)), shape=((M0 + len(m)) * d, (n0 + n1) * d)).tocsr()
        self.b = np.zeros((self.B.shape[0], d))
        self.G = G
        if a == 0:   # node 0 pins its first block (DChordalProblem.cpp:60-62)
            self.g_ = G[d:, :d].toarray()
            self.LG = _spd_solve(G[d:, d:])
        else:
            self.g_ = np.zeros((n0 * d, d))
            self.LG = _spd_solve(G)

    def evaluate_g(self, Y):
        d = self.d
        return self.g_ + (self.S[d:] @ Y if self.a == 0 else self.S @ Y)

    def solve(self, Y, g):
        d = self.d
        if self.a == 0:
            return np.vstack([Y[:d], -self.LG(g)])
        return -self.LG(g)


class ChordalT(_Nesterov):
    """DChordal_t (DChordal_utils.cpp:915-1204, DChordalProblem.cpp:79-104)."""

    def __init__(self, a, meas, xi=REG_G):
        self.a, self.info = a, generate_data_info(a, meas)
        self.d = meas.d
        self.p, self.n_own = 1, self.info.n[0]
        self.xi = xi

    def setup(self, R):
        info, d, a = self.info, self.d, self.a
        n0, n1 = info.n
        g = np.zeros((n0, d))
        rG, cG, vG, rS, cS, vS, rB, cB, vB = ([] for _ in range(9))
        bl = []
        m = info.intra
        for e in range(len(m)):
            i, j = info.index[a][int(m.ipose[e])][1], info.index[a][int(m.jpose[e])][1]
            tau = m.tau[e]
            nt = R[i * d:i * d + d].T @ m.t[e]
            rG += [i, j, i, j]; cG += [i, j, j, i]; vG += [tau, tau, -tau, -tau]
            g[i] += tau * nt
            g[j] -= tau * nt
            rB += [e, e]; cB += [i, j]; vB += [math.sqrt(tau), -math.sqrt(tau)]
            bl.append(math.sqrt(tau) * nt)
        m, M0 = info.inter, len(info.intra)
        for e in range(len(m)):
            (bi, ki), (bj, kj), tail_local = _edge_slots(info, a, e, d)
            ci, cj = info.s[bi] + ki, info.s[bj] + kj
            tau = m.tau[e]
            nt = R[ci * d:ci * d + d].T @ m.t[e]
            if tail_local:
                rG.append(ci); cG.append(ci); vG.append(2 * tau)
                rS += [ci, ci]; cS += [ci, cj]; vS += [-tau, -tau]
                g[ci] += tau * nt
            else:
                rG.append(cj); cG.append(cj); vG.append(2 * tau)
                rS += [cj, cj]; cS += [cj, ci]; vS += [-tau, -tau]
                g[cj] -= tau * nt
            rB += [M0 + e, M0 + e]; cB += [ci, cj]; vB += [math.sqrt(tau), -math.sqrt(tau)]
            bl.append(math.sqrt(tau) * nt)
        for i in range(n0):
            rG.append(i); cG.append(i); vG.append(self.xi)
            rS.append(i); cS.append(i); vS.append(-self.xi)
        self.G = sp.coo_matrix((vG, (rG, cG)), shape=(n0, n0)).tocsc()
        self.S = sp.coo_matrix((vS, (rS, cS)), shape=(n0, n0 + n1)).tocsr()
        self.B = sp.coo_matrix((vB, (rB, cB)), shape=(M0 + len(m), n0 + n1)).tocsr()
        self.b = np.array(bl).reshape(-1, d)
        self.g_ = g
        self.LG = _spd_solve(self.G)

    def evaluate_g(self, Y):
        return self.g_ + self.S @ Y

    def solve(self, Y, g):
        return -self.LG(g)


# ---------------------------------------------------------------------------------------------------------
def _communicate(infos, xs, p):
    """DChordal::communicate (DChordal_utils.h:196-240): grow xs[a] to p (n0 + n1) rows and fill the neighbour rows."""
    out = []
    for a, info in enumerate(infos):
        n0, n1 = info.n
        Z = np.zeros((p * (n0 + n1), xs[a].shape[1]))
        Z[:p * n0] = xs[a][:p * n0]
        out.append(Z)
    for a, info in enumerate(infos):
        n0 = info.n[0]
        for b, poses in info.index.items():
            if b == a:
                continue
            for j, (_, k) in poses.items():
                out[a][p * (n0 + k): p * (n0 + k) + p] = xs[b][p * j: p * j + p]
    return out


def _communicate_poses(infos, xs, d):
    """DPGO::communicate (DPGO_utils.h:397-453) on [t ; R] matrices: grow to (d+1)(n0+n1) rows, fill neighbours."""
    out = []
    for a, info in enumerate(infos):
        n0, n1 = info.n
        Z = np.zeros(((d + 1) * (n0 + n1), d))
        Z[:(d + 1) * n0] = xs[a][:(d + 1) * n0]
        out.append(Z)
    for a, info in enumerate(infos):
        n0, n1 = info.n
        for b, poses in info.index.items():
            if b == a:
                continue
            nb0 = infos[b].n[0]
            for j, (_, k) in poses.items():
                out[a][(d + 1) * n0 + k] = xs[b][j]
                r0 = (d + 1) * n0 + n1 + k * d
                out[a][r0:r0 + d] = xs[b][nb0 + j * d: nb0 + j * d + d]
    return out


def _n_communicate(stages):
    """DChordalReduced::n_communicate (DChordalReduced.h:24-51)."""
    for st in stages:
        for b, i in st.n_index.items():
            if b != st.a:
                st.Xk[i * st.p:(i + 1) * st.p] = stages[b].Xk[:st.p]


def _chordal_communicate(stages):
    """DChordal::communicate (DChordal.h:26-84)."""
    for st in stages:
        n0, p = st.info.n[0], st.p
        for b, poses in st.info.index.items():
            if b == st.a:
                continue
            for j, (_, k) in poses.items():
                st.Xk[p * (n0 + k): p * (n0 + k) + p] = stages[b].Xk[p * j: p * j + p]


def local_solve(a, meas, iters=LOCAL_ITERS):
    """Stage-0 stand-in (see the module docstring): poses of node a from its intra-node edges only, as X = [t ; R^T
    blocks] ((d+1) n0 x d), before the gauge change of dist_pgo.cpp:156-157."""
    from .hash import DPGOHash, Options, SCHEME_MM
    from .star import chordal_initialization
    intra = meas.take(np.nonzero((meas.inode == a) & (meas.jnode == a))[0])
    n0 = int(max(intra.ipose.max(), intra.jpose.max())) + 1
    z = np.zeros(len(intra), np.int64)
    X0 = chordal_initialization(n0, Measurements(z, intra.ipose, z, intra.jpose, intra.R, intra.t, intra.kappa, intra.tau))
    o = Options.driver(0, False)
    o.scheme = SCHEME_MM
    o.accepted_delta = 0.0        # refinement on in every iteration
    nd = DPGOHash(a, intra, o)
    nd.initialize(X0)
    nd.update()
    for _ in range(iters):
        nd.iterate()
        nd.update()
    return nd.results.Xk.copy()


def dist_chordal_initialization(measurements, local_solutions=None, schedule=SCHEDULE, trace=None):
    """dist_pgo.cpp:144-416.  measurements[a]: every edge touching node a (read_g2o).  Returns Xk[a] =
    [t (n0 x d) ; R^T blocks (d n0 x d)] per node.  trace (a dict) receives the per-stage objectives
    0.5 * sum_a |B X + b|^2 sampled every 20 iterations, and the stage outputs."""
    N = len(measurements)
    d = measurements[0].d
    infos = [generate_data_info(a, measurements[a]) for a in range(N)]
    tr = trace if trace is not None else {}
    # ---- stage 0: local solutions in the gauge where the first rotation is the identity (:147-158)
    xs = []
    for a in range(N):
        X = local_solve(a, measurements[a]) if local_solutions is None else np.asarray(local_solutions[a])
        n0 = infos[a].n[0]
        xs.append(X[:(d + 1) * n0] @ X[n0:n0 + d])          # xhat^T xhat[:, n:n+d]
    tr["stage0"] = [x.copy() for x in xs]
    # ---- stage 1: reduced rotations (:160-225)
    xs = _communicate_poses(infos, xs, d)
    red_R = [ReducedR(a, measurements[a]) for a in range(N)]
    for a, st in enumerate(red_R):
        st.setup(xs[a])
        st.initialize(np.tile(np.eye(d), (st.nn + 1, 1)))
    obj = []
    for it in range(schedule[0]):
        if it % 20 == 0:
            obj.append(0.5 * sum(st.objective() for st in red_R))
        for st in red_R[1:]:
            st.update()
            st.iterate()
        _n_communicate(red_R)
    tr["objective_reduced_R"] = obj
    rots_n = [project_to_SOd(st.Xak) for st in red_R]
    tr["rots_n"] = [r.copy() for r in rots_n]
    # ---- stage 2: rotations (:230-304)
    stR = [ChordalR(a, measurements[a]) for a in range(N)]
    rots = [xs[a][infos[a].n[0]:(d + 1) * infos[a].n[0]] @ rots_n[a] for a in range(N)]
    rots = _communicate(infos, rots, d)
    for a, st in enumerate(stR):
        st.setup()
        st.initialize(rots[a])
    obj = []
    for it in range(schedule[1]):
        if it % 20 == 0:
            obj.append(0.5 * sum(st.objective() for st in stR))
        for st in stR:
            st.update()
            st.iterate()
        _chordal_communicate(stR)
    tr["objective_R"] = obj
    rots = [project_to_SOdn(st.Xak, d) for st in stR]
    rots_n = [r[:d].copy() for r in rots]
    for a in range(N):
        n0 = infos[a].n[0]
        xs[a][n0:(d + 1) * n0] = rots[a][:d * n0] @ rots_n[a].T
    tr["rots"] = [r.copy() for r in rots]
    # ---- stage 3: reduced translations (:311-359)
    red_t = [ReducedT(a, measurements[a]) for a in range(N)]
    for a, st in enumerate(red_t):
        n0 = infos[a].n[0]
        xs[a][:n0] = st.recover_translations(xs[a][n0:(d + 1) * n0])
    xs = _communicate_poses(infos, xs, d)
    nRs = []
    for a, st in enumerate(red_R):       # DChordal::n_communicate(problems_red_R, rots_n), DChordal_utils.h:148-190
        Z = np.zeros((d * (st.nn + 1), d))
        Z[:d] = rots_n[a]
        for b, i in st.n_index.items():
            if b != a:
                Z[d * i:d * i + d] = rots_n[b][:d]
        nRs.append(Z)
    for a, st in enumerate(red_t):
        st.setup(xs[a], nRs[a])
        st.initialize(np.zeros((st.nn + 1, d)))
    obj = []
    for it in range(schedule[2]):
        if it % 20 == 0:
            obj.append(0.5 * sum(st.objective() for st in red_t))
        for st in red_t[1:]:
            st.update()
            st.iterate()
        _n_communicate(red_t)
    tr["objective_reduced_t"] = obj
    # ---- stage 4: translations (:365-407)
    ts = [xs[a][:infos[a].n[0]] @ nRs[a][:d] + red_t[a].Xak[0] for a in range(N)]
    rots = _communicate(infos, rots, d)
    ts = _communicate(infos, ts, 1)
    stT = [ChordalT(a, measurements[a]) for a in range(N)]
    for a, st in enumerate(stT):
        st.setup(rots[a])
        st.initialize(ts[a])
    obj = []
    for it in range(schedule[3]):
        if it % 20 == 0:
            obj.append(0.5 * sum(st.objective() for st in stT))
        for st in stT:
            st.update()
            st.iterate()
        _chordal_communicate(stT)
    tr["objective_t"] = obj
    Xk = []
    for a in range(N):
        n0 = infos[a].n[0]
        Xk.append(np.vstack([stT[a].Xak, rots[a][:d * n0]]))      # :409-415
    return Xk
