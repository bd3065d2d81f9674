"""Synthetic SE(3) lattice pose graphs (BASELINE.json config 4).

Lattice nx x ny x nz, pose id = x + nx*y + nx*ny*z (z-major, so the reference's contiguous
partition gives every node a slab of whole z-layers).  Edges: all 6-neighbour lattice edges plus
loop closures between lattice points at Chebyshev distance <= 2, drawn without replacement, written
i < j.  Ground truth t = (x, y, z), R = I; translation noise N(0, 0.1^2 I), rotation noise
exp(N(0, 0.05^2 I)); information diag(100,100,100,400,400,400) => tau = 100, kappa = 200
(C++/DPGO/src/DPGO_utils.cpp:107-116); `outlier_frac` of the closures are replaced by uniformly
random rotations / translations inside the bounding box.

The headline instance is grid(50, 50, 40, 400_000): 100 000 poses, 293 500 lattice edges + 106 500
closures = 400 000 edges, seed 20240817 (numpy default_rng -- this module defines the instance).
Data generation only; nothing here is on the timed path.
"""
from __future__ import annotations

import numpy as np

HEADLINE = dict(nx=50, ny=50, nz=40, num_edges=400_000, seed=20240817)


def _exp_so3(w):
    th = np.linalg.norm(w, axis=1)
    k = w / np.maximum(th, 1e-300)[:, None]
    K = np.zeros((len(w), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    s, c = np.sin(th)[:, None, None], np.cos(th)[:, None, None]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def _random_rotations(rng, n):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    w, x, y, z = q.T
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def grid(nx, ny, nz, num_edges=None, seed=20240817, outlier_frac=0.02, sigma_t=0.1, sigma_r=0.05,
         tau=100.0, kappa=200.0):
    """Returns dict(d, num_poses, I, J, R, t, kappa, tau, outlier) for graph_from_edges / the oracle."""
    rng = np.random.default_rng(seed)
    N = nx * ny * nz
    x, y, z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    pid = (x + nx * y + nx * ny * z)
    lat = []
    lat.append((pid[:-1].ravel(), pid[1:].ravel()))
    lat.append((pid[:, :-1].ravel(), pid[:, 1:].ravel()))
    lat.append((pid[:, :, :-1].ravel(), pid[:, :, 1:].ravel()))
    I = np.concatenate([a for a, _ in lat])
    J = np.concatenate([b for _, b in lat])
    n_lat = len(I)
    n_extra = 0 if num_edges is None else num_edges - n_lat
    assert n_extra >= 0
    seen = set((I * N + J).tolist())
    ei, ej = [], []
    while len(ei) < n_extra:
        k = max(2 * (n_extra - len(ei)), 1024)
        a = rng.integers(0, N, k)
        off = rng.integers(-2, 3, (k, 3))
        ax, ay, az = a % nx, (a // nx) % ny, a // (nx * ny)
        bx, by, bz = ax + off[:, 0], ay + off[:, 1], az + off[:, 2]
        ok = (bx >= 0) & (bx < nx) & (by >= 0) & (by < ny) & (bz >= 0) & (bz < nz) & (np.abs(off).sum(1) > 0)
        b = bx + nx * by + nx * ny * bz
        for u, v in zip(a[ok].tolist(), b[ok].tolist()):
            lo, hi = (u, v) if u < v else (v, u)
            key = lo * N + hi
            if key in seen:
                continue
            seen.add(key)
            ei.append(lo)
            ej.append(hi)
            if len(ei) == n_extra:
                break
    I = np.concatenate([I, np.asarray(ei, np.int64)]).astype(np.int64)
    J = np.concatenate([J, np.asarray(ej, np.int64)]).astype(np.int64)
    M = len(I)
    pos = np.stack([I % nx, (I // nx) % ny, I // (nx * ny)], 1).astype(float)
    posj = np.stack([J % nx, (J // nx) % ny, J // (nx * ny)], 1).astype(float)
    R = _exp_so3(sigma_r * rng.standard_normal((M, 3)))          # R_i = R_j = I
    t = (posj - pos) + sigma_t * rng.standard_normal((M, 3))
    outlier = np.zeros(M, bool)
    if n_extra and outlier_frac > 0:
        n_out = int(round(outlier_frac * n_extra))
        sel = n_lat + rng.choice(n_extra, n_out, replace=False)
        outlier[sel] = True
        R[sel] = _random_rotations(rng, n_out)
        t[sel] = rng.uniform(-1, 1, (n_out, 3)) * np.array([nx, ny, nz])
    return dict(d=3, num_poses=N, I=I, J=J, R=R, t=t, kappa=np.full(M, kappa), tau=np.full(M, tau),
                outlier=outlier)


def write_g2o(path, g):
    """EDGE_SE3:QUAT writer so that the same instance can feed any g2o reader."""
    with open(path, "w") as fh:
        for e in range(len(g["I"])):
            R = g["R"][e]
            w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
            if w > 1e-6:
                q = [(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w]
            else:
                ev, evec = np.linalg.eigh((R + R.T) / 2)
                ax = evec[:, -1]
                q = [ax[0], ax[1], ax[2], 0.0]
            ti, ki = g["tau"][e], 2 * g["kappa"][e]
            info = "%g 0 0 0 0 0 %g 0 0 0 0 %g 0 0 0 %g 0 0 %g 0 %g" % (ti, ti, ti, ki, ki, ki)
            fh.write("EDGE_SE3:QUAT %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %s\n" % (
                g["I"][e], g["J"][e], *g["t"][e], *q, info))
