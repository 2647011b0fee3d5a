#!/bin/bash
# TLPK_CHAIN_MIN_NS again, now that a block column of the chain costs 110 us instead of 200: 257 / 513 / 769 (default) / 1025 on the latency-bound LPs, rank-local N = 8, C4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06aq
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
for rep in 1 2; do
for mn in 769 513 257 1025; do
  for wl in pds stair25; do
    TLPK_CHAIN_MIN_NS=$mn timeout 300 python bench.py --workload $wl $S > ${O}_b.json 2> ${O}_b.err
    python - <<P
import json
d=json.load(open("${O}_b.json")); print("min_ns=$mn $wl", round(d["ms_per_step"],4), d.get("ms_per_step_runs"))
P
  done
done
done
for mn in 769 513 257; do TLPK_CHAIN_MIN_NS=$mn NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "nranks" | cut -c1-140; done
for mn in 769 257; do TLPK_CHAIN_MIN_NS=$mn timeout 300 python bench.py --workload c4 $S > ${O}_b.json 2> ${O}_b.err; python -c "
import json; d=json.load(open('${O}_b.json')); print('min_ns=$mn c4', round(d['ms_per_step'],3))"; done
