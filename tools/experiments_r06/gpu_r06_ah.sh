#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06ah
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_hsd_device.py tests/test_lp_configs.py tests/test_presolve.py -m gpu -x -q 2>&1 | tail -3
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { local name=$1; local wl=$2; shift; shift
    env "$@" timeout 600 python bench.py --workload $wl $S > ${O}_bench_${wl}_$name.json 2> ${O}_bench_${wl}_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_${wl}_$name.json")); print("$wl $name", round(d["ms_per_step"],4), d.get("ms_per_step_runs"), "launches", d["config"].get("launches_update"), d["config"].get("launches_solve"))
P
}
for rep in 1 2; do
run merge stair25 X=1
run nomerge stair25 TLPK_SOLVE_MERGE=0
run merge pds X=1
run nomerge pds TLPK_SOLVE_MERGE=0
run merge c4 X=1
run nomerge c4 TLPK_SOLVE_MERGE=0
done
