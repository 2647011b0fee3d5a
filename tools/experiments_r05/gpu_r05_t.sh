#!/bin/bash
# Round 5, GPU session T: TLPK_CHAIN_CUS = R -- R compute units reserved for the diagonal-block chains (own CU-masked stream; every other stream masked off them):
# pds-class LP (its chain does not overlap the rows-below update: profiles/r05_timeline_pds.txt), 25fv47-class, rank-local N = 8, C4; parity subset with the reservation on.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05t
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.2f ms  runs %s" % (d["ms_per_step"], d["ms_per_step_runs"]))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi --no-roofline"
for r in 0 2 4 8 0 4; do
  for wl in pds stair25 c4; do echo "$wl TLPK_CHAIN_CUS=$r: $(TLPK_CHAIN_CUS=$r timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "$show")"; done
  echo "rank-local TLPK_CHAIN_CUS=$r: $(TLPK_CHAIN_CUS=$r NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1 | cut -c1-130)"
done | tee ${O}_chain_cus.txt
echo "pds TLPK_GRAPH=0 TLPK_CHAIN_CUS=0: $(TLPK_GRAPH=0 timeout 300 python bench.py --workload pds $B 2>/dev/null | python -c "$show")" | tee -a ${O}_chain_cus.txt
S="--workload pds --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
TLPK_CHAIN_CUS=4 timeout 400 rocprofv3 --kernel-trace --output-format csv -d ${O}_trace -- python bench.py $S > ${O}_trace.log 2>&1
STEP=4 python tools/timeline_overlap.py $(ls ${O}_trace/*/*kernel_trace.csv | head -1) > ${O}_timeline_pds_chain4.txt 2>&1; tail -14 ${O}_timeline_pds_chain4.txt; rm -rf ${O}_trace
TLPK_CHAIN_CUS=4 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -q -x 2>&1 | tail -3 | tee ${O}_pytest.txt
