#!/bin/bash
# Round 5, GPU session Q: with the faster diagonal blocks -- (1) the look-ahead rule again (auto / 0 / 1) on the pds-class LP and on rank-local N = 8,
# (2) tlpk_update_device_async (the root front's chain under the first solve's block-level sweeps; slower in round 3), (3) stream groups 1 / 2 / 3 at N = 8.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05q
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernel_ms",{})
print("%.2f ms  runs %s  potrf %s trsm %s update %s" % (d["ms_per_step"], d["ms_per_step_runs"], k.get("potrf"), k.get("trsm"), k.get("update")))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi"
for la in auto 0 1; do
  if [ $la = auto ]; then unset TLPK_LOOKAHEAD; else export TLPK_LOOKAHEAD=$la; fi
  echo "pds lookahead=$la: $(timeout 300 python bench.py --workload pds $B 2>/dev/null | python -c "$show")"
  echo "rank-local lookahead=$la: $(NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1 | cut -c1-260)"
done | tee ${O}_lookahead.txt
unset TLPK_LOOKAHEAD
for a in "" "--async-update" "" "--async-update"; do echo "c4 [$a]: $(timeout 300 python bench.py $B $a 2>/dev/null | python -c "$show")"; done | tee ${O}_async.txt
for ng in 1 2 3; do echo "rank-local NG=$ng: $(NG=$ng NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1 | cut -c1-260)"; done | tee ${O}_groups.txt
