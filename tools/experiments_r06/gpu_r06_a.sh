#!/bin/bash
# round 6, session A: first run of the dependency-driven factorisation on the device
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06a
timeout 900 python -m pytest tests/test_chain.py -m gpu -x -q > ${O}_pytest_chain.txt 2>&1
tail -5 ${O}_pytest_chain.txt
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
for ch in 0 -1; do
  for wl in pds stair25 c4; do
    if [ $ch = 0 ]; then export TLPK_CHAIN=0; else unset TLPK_CHAIN; fi
    timeout 300 python bench.py --workload $wl $S > ${O}_bench_${wl}_chain${ch}.json 2> ${O}_bench_${wl}_chain${ch}.err
    python - <<P
import json
try:
    d=json.load(open("${O}_bench_${wl}_chain${ch}.json")); print("$wl chain=$ch", d["ms_per_step"], d.get("ms_per_step_runs"))
except Exception as e: print("$wl chain=$ch failed", e)
P
  done
done
