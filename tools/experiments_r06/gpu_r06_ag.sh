#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06ag
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { local name=$1; local wl=$2; shift; shift
    env "$@" timeout 600 python bench.py --workload $wl $S > ${O}_bench_${wl}_$name.json 2> ${O}_bench_${wl}_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_${wl}_$name.json")); print("$wl $name", round(d["ms_per_step"],4), d.get("ms_per_step_runs"), "launches", d["config"].get("launches_update"), d["config"].get("launches_solve"))
P
}
for rep in 1 2; do
run default stair25 X=1
run serial stair25 TLPK_SERIAL=1
run nograph stair25 TLPK_GRAPH=0
run default pds X=1
run serial pds TLPK_SERIAL=1
run nograph pds TLPK_GRAPH=0
done
