#!/usr/bin/env python3
"""HIP API calls per outer iteration from two rocprofv3 --hip-runtime-trace runs of the same command that differ only in the
number of steps: (calls of the long run - calls of the short run) / (difference in steps).  Set-up, warm-up and graph
captures cancel.  Usage: api_per_iteration.py <dir_short> <steps_short> <dir_long> <steps_long>"""
import collections
import csv
import glob
import sys


def counts(d):
    f = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
    return collections.Counter(r["Function"] for r in csv.DictReader(open(f)))


a, na, b, nb = counts(sys.argv[1]), int(sys.argv[2]), counts(sys.argv[3]), int(sys.argv[4])
SUBMIT = ("hipLaunchKernel", "hipExtLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel", "hipGraphLaunch", "hipMemcpyAsync",
          "hipMemsetAsync", "hipEventRecord", "hipStreamWaitEvent", "hipLaunchCooperativeKernel")
tot = 0.0
print("%-34s %10s" % ("HIP API call", "per iteration"))
for k in sorted(set(a) | set(b), key=lambda k: -(b[k] - a[k])):
    per = (b[k] - a[k]) / (nb - na)
    if abs(per) < 1e-9:
        continue
    mark = "  <- submission" if k in SUBMIT else ""
    print("%-34s %10.2f%s" % (k, per, mark))
    if k in SUBMIT:
        tot += per
print("%-34s %10.2f" % ("submissions to the GPU per iteration", tot))
