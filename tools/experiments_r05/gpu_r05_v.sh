#!/bin/bash
# Round 5, GPU session V: chain first (TLPK_CHAIN_FIRST=1) with the chain's waves at raised priority (TLPK_POTRF_PRIO=1: s_setprio 3 in k_potrf_wide) -- does the overlap pay
# when the chain is not starved by the update's waves on its SIMDs?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05v
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.2f ms  runs %s" % (d["ms_per_step"], d["ms_per_step_runs"]))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi --no-roofline"
for v in "0 0" "1 1" "0 1" "1 0" "0 0" "1 1"; do
  set -- $v; export TLPK_CHAIN_FIRST=$1 TLPK_POTRF_PRIO=$2
  for wl in pds c4; do echo "$wl chain_first=$1 prio=$2: $(timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "$show")"; done
  echo "rank-local chain_first=$1 prio=$2: $(NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1 | cut -c1-130)"
done | tee ${O}_prio.txt
export TLPK_CHAIN_FIRST=1 TLPK_POTRF_PRIO=1
S="--workload pds --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
TLPK_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d ${O}_trace -- python bench.py $S > ${O}_trace.log 2>&1
STEP=4 python tools/timeline_overlap.py $(ls ${O}_trace/*/*kernel_trace.csv | head -1) > ${O}_timeline_pds.txt 2>&1; tail -13 ${O}_timeline_pds.txt; rm -rf ${O}_trace
