#!/bin/bash
# Run-to-run reproducibility of a long run with a host that comes late to its read-backs by 0 / 30 / 200 us: the results
# must not depend on how far the stream runs ahead of the host.  Usage: tools/r6_repro.sh [ENV=VALUE ...]
export DPGO_ITER_GRAPH=${DPGO_ITER_GRAPH:-0}
python tools/probes/repro_run.py /tmp/warm.npy 2>/dev/null   # (the first run of a fresh box: discarded)
run() { tag=$1; shift; n=0; for d in 0 30 200 0; do env DPGO_DEBUG_LATE_HOST_US=$d "$@" python tools/probes/repro_run.py /tmp/$tag.$n.npy 2>/dev/null; n=$((n+1)); done; python tools/probes/repro_cmp.py $tag /tmp/$tag.0.npy /tmp/$tag.1.npy /tmp/$tag.2.npy /tmp/$tag.3.npy; }
run base A=1
run nocggraph DPGO_CG_GRAPH=0
run norefine DPGO_SPEC_REFINE=0
run norefine_nocg DPGO_SPEC_REFINE=0 DPGO_CG_GRAPH=0
run unfused DPGO_FUSED=0 DPGO_SPEC_UPDATE=0 DPGO_LAZY_UPDATE_REDUCE=0 DPGO_LAZY_UNPACK=0
run alloff DPGO_FUSED=0 DPGO_SPEC_UPDATE=0 DPGO_LAZY_UPDATE_REDUCE=0 DPGO_LAZY_UNPACK=0 DPGO_SPEC_REFINE=0 DPGO_CG_GRAPH=0 DPGO_DEFER_UPDATE=0
python tools/probes/repro_cmp.py base_vs_nocggraph /tmp/base.0.npy /tmp/nocggraph.0.npy
python tools/probes/repro_cmp.py base_vs_unfused /tmp/base.0.npy /tmp/unfused.0.npy
python tools/probes/repro_cmp.py base_vs_alloff /tmp/base.0.npy /tmp/alloff.0.npy
