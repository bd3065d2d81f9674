#!/usr/bin/env python3
"""Writes tests/golden/so_ref.npz: seeded 3x3 / 2x2 inputs and the outputs of the REFERENCE's AVX2 nearest-rotation
kernels (oracle/_ref/so_ref, built by oracle/ref_so3/Makefile from C++/DPGO/src/internal/project_to_SOd.cpp where it
lies).  Run in the container that has /root/reference; the vectors (data, not source) are committed."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "oracle", "_ref", "so_ref")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "ref_so3")])


def run(d, A):
    inp = struct.pack("<q", len(A)) + np.ascontiguousarray(A, "<f8").tobytes()
    out = subprocess.run([EXE, str(d)], input=inp, capture_output=True, check=True).stdout
    return np.frombuffer(out, "<f8").reshape(len(A), d, d).copy()


rng = np.random.default_rng(20240817)
n = 1200
A3 = rng.standard_normal((n, 3, 3))
A3[:100] *= 1e-3                                                  # small
A3[100:200] *= 1e3                                                # large
q, _ = np.linalg.qr(rng.standard_normal((400, 3, 3)))
A3[200:600] = q + 1e-6 * rng.standard_normal((400, 3, 3))        # near rotations / near reflections
A3[600:700] = q[:100] * np.array([3.0, 2.0, 1e-9])               # nearly rank 2
A3[700:800] = q[100:200] @ (np.eye(3) * np.array([1.0, 0.5, 0.25])) @ q[200:300]   # generic, well conditioned
A3[800] = 2 * np.eye(3)
A3[801] = np.diag([1.0, 1.0, -1.0])
A3[802] = np.zeros((3, 3))
A2 = rng.standard_normal((400, 2, 2))
A2[:50] *= 1e-3
A2[50] = 0.0                                                      # below the 1e-32 guard (traits.cpp:10) -> identity
A2[51] = 1e-17 * np.array([[1.0, 2.0], [-2.0, 1.0]])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "so_ref.npz"), A3=A3, U3=run(3, A3), A2=A2, U2=run(2, A2))
print("written", n, len(A2))
