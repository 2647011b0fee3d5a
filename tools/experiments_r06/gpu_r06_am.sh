#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06am
timeout 1200 python -m pytest tests/test_chain.py tests/test_gpu_parity.py tests/test_k2.py -m gpu -x -q 2>&1 | tail -3
S="--steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { local name=$1; local wl=$2; shift; shift
    env "$@" timeout 600 python bench.py --workload $wl $S > ${O}_bench_${wl}_$name.json 2> ${O}_bench_${wl}_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_${wl}_$name.json")); km=d.get("kernel_ms") or {}
print("$wl $name", round(d["ms_per_step"],3), d.get("ms_per_step_runs"), "trsm", km.get("trsm"), "update", km.get("update"), "chain", km.get("chain"))
P
}
for rep in 1 2; do
run new c4 X=1
run new headline X=1
run new pds X=1
done
timeout 300 python tools/chain_trace.py 2>&1 | grep -E "trsm  |k0= 7" | cut -c1-220
