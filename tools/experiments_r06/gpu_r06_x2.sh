#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_chain.py tests/test_gpu_parity.py tests/test_k2.py tests/test_hsd_device.py -m gpu -x -q 2>&1 | tail -3
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
for wl in pds c4; do
    timeout 300 python bench.py --workload $wl $S > gpurun_out/r06x_bench_${wl}.json 2> gpurun_out/r06x_bench_${wl}.err
    python - <<P
import json
d=json.load(open("gpurun_out/r06x_bench_${wl}.json")); print("$wl", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
done
NLIST=4,8 timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "nranks" | cut -c1-300
