#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06j
timeout 900 python -m pytest tests/test_chain.py tests/test_gpu_parity.py -m gpu -x -q -k "chain or block_kernels or large_fronts or general_sparse or c3_shape or bitwise" 2>&1 | tail -2
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
for wl in pds c4; do
    timeout 300 python bench.py --workload $wl $S > ${O}_bench_${wl}.json 2> ${O}_bench_${wl}.err
    python - <<P
import json
d=json.load(open("${O}_bench_${wl}.json")); print("$wl", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
done
NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "nranks" | cut -c1-300
timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds.txt 2>&1
tail -5 ${O}_chain_trace_pds.txt | cut -c1-250
