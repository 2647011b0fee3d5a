#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06c
timeout 600 python -m pytest tests/test_chain.py -m gpu -x -q 2>&1 | tail -3
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { # name, env...
  name=$1; shift
  for wl in pds stair25; do
    env "$@" timeout 300 python bench.py --workload $wl $S > ${O}_bench_${wl}_${name}.json 2> ${O}_bench_${wl}_${name}.err
    python - <<P
import json
try:
    d=json.load(open("${O}_bench_${wl}_${name}.json")); print("$wl $name", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
except Exception as e: print("$wl $name failed", e)
P
  done
}
run launch TLPK_CHAIN=0
run chain TLPK_X=1
run chain_1wg TLPK_CHAIN_DYNLDS=70000 TLPK_CHAIN_GRID=256
run chain_grid1024 TLPK_CHAIN_GRID=1024
TLPK_CHAIN_DYNLDS=70000 TLPK_CHAIN_GRID=256 timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds_1wg.txt 2>&1
timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds.txt 2>&1
tail -34 ${O}_chain_trace_pds.txt | cut -c1-230
echo; tail -34 ${O}_chain_trace_pds_1wg.txt | cut -c1-230
