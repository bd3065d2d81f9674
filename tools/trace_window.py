#!/usr/bin/env python3
"""A window of a rocprofv3 --kernel-trace run by kernel name: dispatches [a, b) given as fractions of the run; per name the
launches, mean duration, mean gap in front, share of the window's wall time.  Usage: trace_window.py <dir> <from> <to>"""
import csv, glob, sys
from collections import defaultdict
d, fa, fb = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
S = [int(r["Start_Timestamp"]) for r in rows]
E = [int(r["End_Timestamp"]) for r in rows]
def nm(r):
    k = r["Kernel_Name"].split("(dpgo")[0].split("::")[-1]
    base = k.split("<")[0].split("(")[0].strip()
    if base == "k_spd_level":   # dof (translations / rotations) and mode tell the solves apart
        a = k.split("<")[1].split(",")
        return "k_spd_level<dof %s, mode %s>" % (a[1].strip(), a[3].strip())
    if base in ("k_bsr", "k_root_sym", "k_root_combine"):
        return k.split(">")[0][:28] + ">"
    return base[:28]
names = [nm(r) for r in rows]
a, b = int(fa * len(rows)), int(fb * len(rows))
wall = (E[b - 1] - E[a - 1]) / 1e3
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for i in range(max(a, 1), b):
    dur[names[i]] += (E[i] - S[i]) / 1e3
    gap[names[i]] += max(0, S[i] - E[i - 1]) / 1e3
    cnt[names[i]] += 1
print("dispatches %d..%d of %d: %.1f ms wall, %.1f ms busy, %.1f ms gaps" % (a, b, len(rows), wall / 1e3, sum(dur.values()) / 1e3, sum(gap.values()) / 1e3))
for k in sorted(dur, key=lambda k: -(dur[k] + gap[k])):
    print("  %-34s %7d launches  mean %6.2f us  gap in front %6.2f us  %5.1f %% of the window" % (k, cnt[k], dur[k] / cnt[k], gap[k] / cnt[k], 100.0 * (dur[k] + gap[k]) / wall))
