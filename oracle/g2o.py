"""Oracle: .g2o loader, node partition and per-node index maps.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in numpy:
  * read_g2o_file      C++/DPGO/src/DPGO_utils.cpp:8-138
  * read_g2o           C++/DPGO/src/DPGO_utils.cpp:140-202
  * generate_data_info C++/DPGO/src/DPGO_utils.cpp:326-438
"""
from __future__ import annotations

import numpy as np


class Measurements:
    """A list of relative pose measurements in struct-of-arrays form.

    Mirrors ``RelativePoseMeasurement`` (C++/DPGO/include/DPGO/
    RelativePoseMeasurement.h:11-36): tail ``i`` and head ``j`` as
    (node, pose) pairs, rotation R (d x d), translation t (d), kappa, tau.
    """

    def __init__(self, inode, ipose, jnode, jpose, R, t, kappa, tau):
        self.inode = np.asarray(inode, dtype=np.int64)
        self.ipose = np.asarray(ipose, dtype=np.int64)
        self.jnode = np.asarray(jnode, dtype=np.int64)
        self.jpose = np.asarray(jpose, dtype=np.int64)
        self.R = np.asarray(R, dtype=np.float64)
        self.t = np.asarray(t, dtype=np.float64)
        self.kappa = np.asarray(kappa, dtype=np.float64)
        self.tau = np.asarray(tau, dtype=np.float64)

    def __len__(self):
        return len(self.kappa)

    @property
    def d(self):
        return self.t.shape[1]

    def take(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        return Measurements(self.inode[idx], self.ipose[idx], self.jnode[idx],
                            self.jpose[idx], self.R[idx], self.t[idx],
                            self.kappa[idx], self.tau[idx])


def _quat_to_rot(w, x, y, z):
    # Eigen::Quaternion::toRotationMatrix (no normalisation), as used at
    # DPGO_utils.cpp:100-101.
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def read_g2o_file(filename):
    """DPGO_utils.cpp:8-138.  Returns (num_poses, Measurements) with global
    pose ids stored in ipose/jpose and node = 0."""
    I, J, Rs, ts, kap, tau = [], [], [], [], [], []
    d = None
    with open(filename) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "EDGE_SE2":
                i, j = int(tok[1]), int(tok[2])
                dx, dy, dth, I11, I12, I13, I22, I23, I33 = map(float, tok[3:12])
                c, s = np.cos(dth), np.sin(dth)
                Rs.append(np.array([[c, -s], [s, c]]))
                ts.append(np.array([dx, dy]))
                cov = np.array([[I11, I12], [I12, I22]])
                tau.append(2.0 / np.trace(np.linalg.inv(cov)))   # :63-65
                kap.append(I33)                                   # :67
                d = 2
            elif tok[0] == "EDGE_SE3:QUAT":
                i, j = int(tok[1]), int(tok[2])
                v = list(map(float, tok[3:31]))
                dx, dy, dz, qx, qy, qz, qw = v[:7]
                (I11, I12, I13, I14, I15, I16, I22, I23, I24, I25, I26, I33,
                 I34, I35, I36, I44, I45, I46, I55, I56, I66) = v[7:28]
                Rs.append(_quat_to_rot(qw, qx, qy, qz))
                ts.append(np.array([dx, dy, dz]))
                tc = np.array([[I11, I12, I13], [I12, I22, I23], [I13, I23, I33]])
                tau.append(3.0 / np.trace(np.linalg.inv(tc)))     # :107-109
                rc = np.array([[I44, I45, I46], [I45, I55, I56], [I46, I56, I66]])
                kap.append(3.0 / (2.0 * np.trace(np.linalg.inv(rc))))  # :114-116
                d = 3
            elif tok[0] in ("VERTEX_SE2", "VERTEX_SE3:QUAT"):
                continue
            else:
                raise ValueError("unrecognized type: %s" % tok[0])   # :121-124
            I.append(i)
            J.append(j)
    num_poses = max(max(I), max(J)) + 1
    z = np.zeros(len(I), dtype=np.int64)
    return num_poses, Measurements(z, I, z, J, np.array(Rs), np.array(ts), kap, tau)


def partition_index(num_poses, num_nodes):
    """The (node, pose) map of DPGO_utils.cpp:147-158 for every global id."""
    q = num_poses // num_nodes
    inc_n = num_poses - num_nodes * q
    inc = inc_n * (q + 1)
    g = np.arange(num_poses)
    node = np.where(g < inc, g // (q + 1), (g - inc) // max(q, 1) + inc_n)
    pose = np.where(g < inc, g % (q + 1), (g - inc) % max(q, 1))
    return node.astype(np.int64), pose.astype(np.int64)


def read_g2o(filename, num_nodes):
    """DPGO_utils.cpp:140-202.

    Returns (num_poses, measurements[node], g_index[node]) where
    measurements[node] holds every edge touching that node (inter-node edges
    appear in both endpoint lists, file order preserved) and g_index[node]
    maps local pose id -> global pose id.
    """
    num_poses, mm = read_g2o_file(filename)
    return partition_measurements(num_poses, mm, num_nodes)


def partition_measurements(num_poses, mm, num_nodes):
    node_of, pose_of = partition_index(num_poses, num_nodes)
    inode, ipose = node_of[mm.ipose], pose_of[mm.ipose]
    jnode, jpose = node_of[mm.jpose], pose_of[mm.jpose]
    g_index = [dict() for _ in range(num_nodes)]
    for e in range(len(mm)):
        g_index[inode[e]].setdefault(int(ipose[e]), int(mm.ipose[e]))
        g_index[jnode[e]].setdefault(int(jpose[e]), int(mm.jpose[e]))
    allm = Measurements(inode, ipose, jnode, jpose, mm.R, mm.t, mm.kappa, mm.tau)
    measurements = []
    for a in range(num_nodes):
        sel = np.nonzero((inode == a) | (jnode == a))[0]
        measurements.append(allm.take(sel))
    return num_poses, measurements, g_index


class DataInfo:
    """Output of generate_data_info (DPGO_utils.cpp:326-438)."""

    def __init__(self):
        self.intra = None       # Measurements
        self.inter = None       # Measurements
        self.n = [0, 0]         # own / neighbour pose counts
        self.s = [0, 0]         # offsets (s[1] = n[0])
        self.m = [0, 0]         # intra / inter measurement counts
        self.index = {}         # node -> {pose -> (0|1, local idx)}
        self.sent = {}          # nbr node -> {own pose -> (0, local idx)}
        self.recv = {}          # nbr node -> {nbr pose -> (1, local idx)}


def generate_data_info(a, meas):
    info = DataInfo()
    if len(meas) == 0:
        raise ValueError("No measurements are specified for node %d" % a)
    is_intra = (meas.inode == a) & (meas.jnode == a)
    info.intra = meas.take(np.nonzero(is_intra)[0])
    info.inter = meas.take(np.nonzero(~is_intra)[0])
    info.m = [len(info.intra), len(info.inter)]
    keys = set(zip(meas.inode.tolist(), meas.ipose.tolist()))
    keys |= set(zip(meas.jnode.tolist(), meas.jpose.tolist()))
    # ordering: own poses by id, then neighbours by (node, id)   :400-418
    own = sorted(p for (nd, p) in keys if nd == a)
    nbr = sorted((nd, p) for (nd, p) in keys if nd != a)
    index = {a: {}}
    for k, p in enumerate(own):
        index[a][p] = (0, k)
    for k, (nd, p) in enumerate(nbr):
        index.setdefault(nd, {})[p] = (1, k)
    info.index = index
    info.n = [len(own), len(nbr)]
    info.s = [0, len(own)]
    sent = {}
    for e in range(len(info.inter)):
        if info.inter.inode[e] != a:
            sent.setdefault(int(info.inter.inode[e]), set()).add(int(info.inter.jpose[e]))
        if info.inter.jnode[e] != a:
            sent.setdefault(int(info.inter.jnode[e]), set()).add(int(info.inter.ipose[e]))
    info.sent = {b: {p: index[a][p] for p in sorted(ps)} for b, ps in sorted(sent.items())}
    info.recv = {b: dict(v) for b, v in index.items() if b != a}
    return info


def local_rows(info, meas, d):
    """Row indices (translation row, first rotation row) of the tail and head
    of every measurement inside the node's Z = [t^a; R^a; t^nbr; R^nbr]
    (DPGO_utils.cpp:1527-1534: s = (d+1)*num_s[blk], n = num_n[blk])."""
    M = len(meas)
    ti = np.empty(M, np.int64)
    ri = np.empty(M, np.int64)
    tj = np.empty(M, np.int64)
    rj = np.empty(M, np.int64)
    blk_i = np.empty(M, np.int64)
    blk_j = np.empty(M, np.int64)
    for e in range(M):
        bi, ki = info.index[int(meas.inode[e])][int(meas.ipose[e])]
        bj, kj = info.index[int(meas.jnode[e])][int(meas.jpose[e])]
        si, ni = (d + 1) * info.s[bi], info.n[bi]
        sj, nj = (d + 1) * info.s[bj], info.n[bj]
        ti[e], ri[e] = si + ki, si + ni + ki * d
        tj[e], rj[e] = sj + kj, sj + nj + kj * d
        blk_i[e], blk_j[e] = bi, bj
    return ti, ri, tj, rj, blk_i, blk_j
