#!/usr/bin/env python3
"""Per-iteration view of a rocprofv3 --kernel-trace run of bench.py: the last N iterations (an iteration ends with the
last kernel of update(): its k_reduce, or -- round 6 -- the k_inter behind the iteration's k_reduce_gate), for every kernel name the launches, the mean duration and
the mean gap in front of it per iteration; totals per iteration.  Usage: trace_iter.py <dir> [N]"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 15
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
S = [int(r["Start_Timestamp"]) for r in rows]
E = [int(r["End_Timestamp"]) for r in rows]
names = [r["Kernel_Name"].split("(dpgo")[0].split("::")[-1].split("<")[0].split("(")[0].strip()[:24] for r in rows]
# iteration boundaries: every k_inter that is followed (within 3 dispatches) by a k_reduce closes an update()
ends = [i for i in range(len(rows)) if names[i] == "k_reduce" and i >= 1 and names[i - 1] in ("k_inter", "k_tangent_full", "k_bdiag_dot", "k_axpby")]
# (round 6: update()'s reduction rides on the next refinement; an iteration then ends with the k_inter behind its k_reduce_gate)
gate = [i for i in range(len(rows)) if names[i] == "k_reduce_gate"]
if len(gate) > len(ends):
    ends = []
    for g in gate:
        j = g + 1
        while j < len(rows) and j < g + 6 and names[j] != "k_inter":
            j += 1
        if j < len(rows) and names[j] == "k_inter":
            ends.append(j)
ends = ends[-(N + 1):]
if len(ends) < 2:
    sys.exit("no iterations found")
a, b = ends[0] + 1, ends[-1] + 1
n = len(ends) - 1
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for i in range(a, b):
    dur[names[i]] += E[i] - S[i]
    gap[names[i]] += max(0, S[i] - E[i - 1])
    cnt[names[i]] += 1
print("%d iterations, %.1f launches / iteration, %.1f us / iteration wall, %.1f us busy, %.1f us gaps" % (
    n, (b - a) / n, (E[b - 1] - E[a - 1]) / 1e3 / n, sum(dur.values()) / 1e3 / n, sum(gap.values()) / 1e3 / n))
for k in sorted(dur, key=lambda k: -dur[k]):
    print("  %-24s %5.1f / iteration  mean %6.2f us  gap in front %6.2f us  total %7.1f us / iteration" % (
        k, cnt[k] / n, dur[k] / 1e3 / cnt[k], gap[k] / 1e3 / cnt[k], (dur[k] + gap[k]) / 1e3 / n))
