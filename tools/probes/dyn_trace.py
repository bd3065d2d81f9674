import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last refactorisation: from the last k_rescale_apply up to the next k_bsr
idx = [i for i, r in enumerate(rows) if "k_rescale_apply" in r["Kernel_Name"]]
i0 = idx[-2]
out = []
prev_end = None
for r in rows[i0:]:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1][:34]
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    out.append((short, (e - s) / 1e3, gap, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]),1), int(r.get("Grid_Size_Y", 1))))
    prev_end = e
    if "k_bsr<" in n and len(out) > 20: break
t0 = None
tot = 0
for o in out:
    print("%-36s dur %7.1f gap %6.1f  grid %5d x %4d" % o)
    tot += o[1] + max(o[2], 0)
print("total", tot)
