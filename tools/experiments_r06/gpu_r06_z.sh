#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06z
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tail_as_64x64 or entrywise or c3_shape" 2>&1 | tail -4
S="--steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { local name=$1; local wl=$2; shift; shift
    env "$@" timeout 600 python bench.py --workload $wl $S > ${O}_bench_${wl}_$name.json 2> ${O}_bench_${wl}_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_${wl}_$name.json")); r=d.get("roofline") or {}
print("$wl $name", round(d["ms_per_step"],3), d.get("ms_per_step_runs"), "roofline frac", round(r.get("frac",0),4), "executed", round(r.get("frac_executed",0),4), "update ms", (d.get("kernel_ms") or {}).get("update"))
P
}
run tail288 c4 X=1
run tail0 c4 TLPK_TAIL64=0
run tail160 c4 TLPK_TAIL64=160
run tail416 c4 TLPK_TAIL64=416
run tail288 headline X=1
run tail0 headline TLPK_TAIL64=0
