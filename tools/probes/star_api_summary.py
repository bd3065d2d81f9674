"""From a rocprofv3 --kernel-trace --hip-runtime-trace run of star_api_trace.py: the HIP API calls and the kernels of the LAST
10 AMM-PGO* iterations (everything after the last hipDeviceSynchronize-free warm-up is hard to mark from Python, so the
window is the last 10/13 of the star kernels): how many host-blocking calls (hipStreamSynchronize, hipDeviceSynchronize,
hipMemcpy, hipEventSynchronize) an iteration makes."""
import csv, glob, sys, collections
d = sys.argv[1]
api = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)
ker = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
K = sorted(csv.DictReader(open(ker[0])), key=lambda r: int(r["Start_Timestamp"]))
star = [r for r in K if "k_star_sums" in r["Kernel_Name"]]
print("k_star_sums launches:", len(star), "(1 at initialisation + 1 per iteration in the common path, + 1 per rare branch)")
t0 = int(star[-10]["Start_Timestamp"]) if len(star) >= 10 else int(star[0]["Start_Timestamp"])
pub = [r for r in K if "k_publish" in r["Kernel_Name"]]
t1 = int(pub[-1]["End_Timestamp"])   # (the window ends with the last iteration's read-back: the process's teardown is not part of it)
A = [r for r in csv.DictReader(open(api[0])) if t0 <= int(r["Start_Timestamp"]) <= t1]
cnt = collections.Counter(r["Function"] for r in A)
print("HIP API calls between the start of the 10th-last k_star_sums and the end of the last k_publish (about 9.5 iterations):")
for k, v in cnt.most_common():
    print("  %-36s %6d" % (k, v))
block = [k for k in cnt if k in ("hipStreamSynchronize", "hipDeviceSynchronize", "hipMemcpy", "hipEventSynchronize", "hipMemcpyAsync")]
print("host-blocking / copy calls in the window:", {k: cnt[k] for k in block} or "none")
def short(n):
    n = n.split("(")[0]
    n = n.split("<")[0]
    return n.split("::")[-1] or n
kc = collections.Counter(short(r["Kernel_Name"]) for r in K if t0 <= int(r["Start_Timestamp"]) <= t1)
print("kernels in the window:", dict(kc.most_common(12)))
