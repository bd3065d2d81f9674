#!/usr/bin/env python3
"""Durations of the device-factorisation kernels (k_fa_*) of a rocprofv3 --kernel-trace run, in launch order, grouped by
consecutive runs of the same kernel.  Usage: fa_trace.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_fa_" in r["Kernel_Name"]]
tot = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split("::")[-1]
    tot[n] = tot.get(n, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print({k: round(v / 1e6, 2) for k, v in tot.items()}, "ms; span %.2f ms" % ((int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6))
abt = [r for r in rows if "k_fa_abt" in r["Kernel_Name"]]
abt.sort(key=lambda r: -(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for r in abt[:25]:
    print("%-40s grid %8s x %6s  %8.1f us" % (r["Kernel_Name"].split("(")[0].split("::")[-1][:40], r["Grid_Size_X"], r["Grid_Size_Y"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
