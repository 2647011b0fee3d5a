#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06l
for n in 256 200 128; do
  echo "== new"; timeout 120 ./tools/potrf_wave_bench 64 $n 2>&1 | cut -c1-400
  echo "== r5";  timeout 120 ./tools/potrf_wave_bench_r5 64 $n 2>&1 | cut -c1-400
done > ${O}_potrf_bench.txt 2>&1
cat ${O}_potrf_bench.txt | cut -c1-330
timeout 900 python -m pytest tests/test_chain.py tests/test_gpu_parity.py tests/test_k2.py -m gpu -x -q 2>&1 | tail -3
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
tail -4 ${O}_chain_trace_pds.txt | cut -c1-250
