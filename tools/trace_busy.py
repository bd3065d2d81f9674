#!/usr/bin/env python3
"""GPU busy fraction of a rocprofv3 --kernel-trace run, window by window: for every window of W consecutive kernel
dispatches, (sum of kernel durations) / (wall span), plus the biggest gaps and which kernel follows them.
Usage: trace_busy.py <dir> [W]"""
import csv
import glob
import sys
from collections import Counter

d = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
S = [int(r["Start_Timestamp"]) for r in rows]
E = [int(r["End_Timestamp"]) for r in rows]
names = [r["Kernel_Name"].split("(")[0].split("::")[-1][:24] for r in rows]
print("%d dispatches, span %.3f s" % (len(rows), (E[-1] - S[0]) / 1e9))
for a in range(0, len(rows), W):
    b = min(a + W, len(rows))
    busy = sum(E[i] - S[i] for i in range(a, b))
    span = E[b - 1] - S[a]
    gaps = Counter()
    gap_ns = 0
    for i in range(a + 1, b):
        g = S[i] - E[i - 1]
        if g > 8000:
            gaps[names[i]] += g
            gap_ns += g
    top = ", ".join("%s %.1f ms" % (k, v / 1e6) for k, v in gaps.most_common(4))
    print("dispatch %7d..%7d  span %8.2f ms  busy %5.1f %%  gaps>8us %6.2f ms  [%s]" % (a, b, span / 1e6, 100.0 * busy / span, gap_ns / 1e6, top))
