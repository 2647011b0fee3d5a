#!/bin/bash
# early entry of the strips (TLPK_CHAIN_EARLY): parity tests, A/B on pds / C4, rank-local N = 8, chain trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06ap
timeout 1500 python -m pytest tests/test_chain.py tests/test_gpu_parity.py tests/test_k2.py tests/test_bump_replay.py -m gpu -x -q 2>&1 | tail -5
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
for rep in 1 2; do
for early in 1 0; do
  for wl in pds c4; do
    TLPK_CHAIN_EARLY=$early timeout 300 python bench.py --workload $wl $S > ${O}_bench_${wl}_${early}.json 2> ${O}_bench_${wl}_${early}.err
    python - <<P
import json
d=json.load(open("${O}_bench_${wl}_${early}.json")); print("early=$early $wl", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
  done
done
done
for early in 1 0; do TLPK_CHAIN_EARLY=$early NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "nranks" | cut -c1-200; done
timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds.txt 2>&1
tail -4 ${O}_chain_trace_pds.txt | cut -c1-250
