#!/usr/bin/env python3
"""Print the kernel timeline (name, grid, duration, gap to the previous kernel) of the last N
dispatches of a rocprofv3 --kernel-trace CSV.  Usage: trace_tail.py <dir> [N]"""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
prev = None
for r in rows:
    name = r["Kernel_Name"].split("(dpgo")[0].split("::")[-1]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-22s grid %8s wg %4s vgpr %3s lds %6s dur %7.1f gap %6.1f" % (
        name[:22], r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["LDS_Block_Size"], (e - s) / 1e3,
        (s - prev) / 1e3 if prev else 0.0))
    prev = e
