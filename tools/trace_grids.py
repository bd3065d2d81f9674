#!/usr/bin/env python3
"""Duration of one kernel of a rocprofv3 --kernel-trace run by grid size (the solve's level launches shrink with the set of live
nodes): dispatches of the kernels whose name contains <pattern> in the window [from, to) of the run (fractions), grouped by
(template arguments, grid): launches, mean / min duration.  Usage: trace_grids.py <dir> <pattern> <from> <to>"""
import csv, glob, sys
from collections import defaultdict
d, pat, fa, fb = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
a, b = int(fa * len(rows)), int(fb * len(rows))
g = defaultdict(list)
for r in rows[a:b]:
    k = r["Kernel_Name"].split("(dpgo")[0].split("::")[-1]
    if pat not in k:
        continue
    g[(k.split(">")[0][:40], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, grid), v in sorted(g.items()):
    if len(v) >= 5:
        print("%-42s %6d workgroups  %6d launches  mean %6.2f us  min %6.2f us" % (k, grid, len(v), sum(v) / len(v), min(v)))
