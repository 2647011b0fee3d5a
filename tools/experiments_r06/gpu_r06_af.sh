#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06af
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_hsd_device.py -m gpu -x -q 2>&1 | tail -3
S="--steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { local name=$1; local wl=$2; shift; shift
    env "$@" timeout 600 python bench.py --workload $wl $S > ${O}_bench_${wl}_$name.json 2> ${O}_bench_${wl}_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_${wl}_$name.json")); sr=d.get("solve_roofline") or {}
print("$wl $name", round(d["ms_per_step"],3), d.get("ms_per_step_runs"), "solve ms", sr.get("ms_per_solve") and round(sr.get("ms_per_solve"),4), "frac", sr.get("frac") and round(sr.get("frac"),4), "pair", (sr.get("pair") or {}).get("ms") and round((sr.get("pair") or {}).get("ms"),4), "unpaired", d.get("unpaired_ms_per_step") and round(d.get("unpaired_ms_per_step"),3))
P
}
for rep in 1 2; do
run side c4 X=1
run noside c4 TLPK_SOLVE_SIDE=0
run side headline X=1
run noside headline TLPK_SOLVE_SIDE=0
run side stair25 X=1
run noside stair25 TLPK_SOLVE_SIDE=0
run side pds X=1
run noside pds TLPK_SOLVE_SIDE=0
done
