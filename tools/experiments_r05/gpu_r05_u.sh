#!/bin/bash
# Round 5, GPU session U: chain first (LK_SIDE_SYNC: the rows-below update of a block column is dispatched together with the diagonal-block kernel, after the diagonal
# tiles) -- TLPK_CHAIN_FIRST unset (rule: levels with 1..16 fronts wider than one block column) / 0 / 1 on the latency-bound LPs, rank-local N = 8, C4, C3; pds timeline; parity.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05u
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.2f ms  runs %s" % (d["ms_per_step"], d["ms_per_step_runs"]))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi --no-roofline"
for cf in auto 0 1 auto 0; do
  if [ $cf = auto ]; then unset TLPK_CHAIN_FIRST; else export TLPK_CHAIN_FIRST=$cf; fi
  for wl in pds stair25 c4; do echo "$wl chain_first=$cf: $(timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "$show")"; done
  echo "rank-local chain_first=$cf: $(NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1 | cut -c1-130)"
done | tee ${O}_chain_first.txt
for cf in auto 0; do
  if [ $cf = auto ]; then unset TLPK_CHAIN_FIRST; else export TLPK_CHAIN_FIRST=$cf; fi
  echo "c3 chain_first=$cf: $(timeout 400 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi --no-roofline 2>/dev/null | python -c "$show")"
done | tee -a ${O}_chain_first.txt
unset TLPK_CHAIN_FIRST
S="--workload pds --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
TLPK_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d ${O}_trace -- python bench.py $S > ${O}_trace.log 2>&1
STEP=4 python tools/timeline_overlap.py $(ls ${O}_trace/*/*kernel_trace.csv | head -1) > ${O}_timeline_pds.txt 2>&1; tail -14 ${O}_timeline_pds.txt; rm -rf ${O}_trace
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_lp_configs.py -m gpu -q 2>&1 | tail -3 | tee ${O}_pytest.txt
