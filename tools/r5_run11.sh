DPGO_ITER_GRAPH=1 timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6
bash tools/final_profile.sh r05 bench rates 2>&1 | tail -30
