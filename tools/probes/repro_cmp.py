"""First iteration at which two or more traces of tools/probes/repro_run.py differ, and in which fields.
Usage: repro_cmp.py <tag> run0.npy run1.npy [...]"""
import sys, numpy as np
names = ["fobj"] * 8 + ["Gk"] * 8 + ["inner"] * 8 + ["restarts"] * 8 + ["X"] * 8
def first(x, y):
    for it in range(len(x)):
        if not np.array_equal(x[it], y[it]):
            d = [f"{names[k]}[{k % 8}]" for k in range(x.shape[1]) if x[it, k] != y[it, k]]
            return it, d
    return None
tag = sys.argv[1]; files = sys.argv[2:]
a = [np.load(f) for f in files]
for i in range(1, len(a)): print(tag, "run0 vs run%d:" % i, first(a[0], a[i]))
