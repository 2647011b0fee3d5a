#!/bin/bash
# Round 5, GPU session J: k_potrf_wide with one trsm call per 64-wide step (TLPK_POTRF_MODE=4) against one call per 64 rows (3, default): microbenchmark, A/B in the library.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05j
for a in "1 256" "64 256" "64 200" "8 600"; do timeout 60 tools/potrf_wave_bench $a | head -1; done 2>&1 | tee ${O}_potrf_bench.txt
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernel_ms",{})
print("%.2f ms  runs %s  potrf %s trsm %s update %s" % (d["ms_per_step"], d["ms_per_step_runs"], k.get("potrf"), k.get("trsm"), k.get("update")))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi"
for m in 3 4 2 3 4; do
  for wl in pds c4; do
    echo "$wl TLPK_POTRF_MODE=$m: $(TLPK_POTRF_MODE=$m timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "$show")"
  done
done | tee ${O}_potrf_ab.txt
for m in 3 2; do echo "rank-local TLPK_POTRF_MODE=$m: $(TLPK_POTRF_MODE=$m timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1)"; done | tee -a ${O}_potrf_ab.txt
