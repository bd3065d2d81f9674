# per-kernel A/B of two library builds on ONE box: .ab/lib_base.so against the current one (bench.py kernels block)
mkdir -p gpurun_out/r4/kab
for rep in 1 2; do for n in base cur; do
  L=$PWD/.ab/lib_$n.so; [ $n = cur ] && L=$PWD/dpgo_amd/libdpgo_amd.so
  DPGO_AMD_LIB=$L timeout 300 python bench.py --no-cpu --traffic off --converge 0 --steps 40 --warmup 10 2>/dev/null > gpurun_out/r4/kab/n1_${n}_$rep.json
  DPGO_AMD_LIB=$L timeout 300 python bench.py --emulate-world 8 --emulate-rank 3 --no-cpu --traffic off --converge 0 --steps 80 --warmup 10 2>/dev/null > gpurun_out/r4/kab/emu_${n}_$rep.json
done; done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r4/kab/*.json")):
    j=json.loads(open(f).read())
    k=j.get("kernels",{})
    print(f.split("/")[-1], "%.4f"%j["ms_per_step"], {n:(round(v.get("avg_us",0),2), v.get("launches_per_step")) for n,v in k.items() if n in ("k_proximal","k_rot_op","k_inter","k_tangent_full")})
PY
