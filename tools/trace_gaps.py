#!/usr/bin/env python3
"""Where the GPU waits for the host: clusters of gaps > T us between consecutive kernels of a rocprofv3 --kernel-trace run
(dispatch index of the first gap of a cluster, gaps in it, their sum, the kernel that follows the first).  Usage: trace_gaps.py <dir> [T]"""
import csv, glob, sys
d = sys.argv[1]
T = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
S = [int(r["Start_Timestamp"]) for r in rows]
E = [int(r["End_Timestamp"]) for r in rows]
names = [r["Kernel_Name"].split("(dpgo")[0].split("::")[-1].split("<")[0].split("(")[0].strip()[:24] for r in rows]
print("%d dispatches" % len(rows))
i = 1
while i < len(rows):
    g = (S[i] - E[i - 1]) / 1e3
    if g > T:
        j, tot, n = i, 0.0, 0
        while j < len(rows) and j < i + 40:
            gj = (S[j] - E[j - 1]) / 1e3
            if gj > T:
                tot += gj; n += 1; last = j
            j += 1
        print("dispatch %7d (%5.1f %% of the run)  %2d gaps > %.0f us within 40 dispatches, %8.1f us in all, first in front of %s" % (i, 100.0 * i / len(rows), n, T, tot, names[i]))
        i = last + 1
    else:
        i += 1
