#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06i
timeout 600 python -m pytest tests/test_chain.py -m gpu -x -q 2>&1 | tail -2
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { name=$1; shift; wls=$1; shift
  for wl in $wls; do
    env "$@" timeout 300 python bench.py --workload $wl $S > ${O}_bench_${wl}_${name}.json 2> ${O}_bench_${wl}_${name}.err
    python - <<P
import json
try:
    d=json.load(open("${O}_bench_${wl}_${name}.json")); print("$wl $name", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
except Exception as e: print("$wl $name failed", e)
P
  done
}
{
run launch "pds" TLPK_CHAIN=0
run chain_tile128 "pds" TLPK_CHAIN_TILE64=0
run chain "pds c4" TLPK_X=1
rl() { name=$1; shift; echo "== $name"; env "$@" timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "nranks" | cut -c1-300; }
rl chain_tile128 NLIST=8 TLPK_CHAIN_TILE64=0
rl chain NLIST=8 TLPK_X=1
} > ${O}_summary.txt 2>&1
cat ${O}_summary.txt
timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds.txt 2>&1
tail -6 ${O}_chain_trace_pds.txt | cut -c1-250
