#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06aj
S="--steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { local name=$1; local wl=$2; shift; shift
    env "$@" timeout 600 python bench.py --workload $wl $S > ${O}_bench_${wl}_$name.json 2> ${O}_bench_${wl}_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_${wl}_$name.json")); sr=d.get("solve_roofline") or {}
print("$wl $name", round(d["ms_per_step"],3), "solve ms", round(sr.get("ms_per_solve"),4), "frac", round(sr.get("frac"),4), "pair", round((sr.get("pair") or {}).get("ms"),4))
P
}
for rep in 1 2; do
run m256 headline X=1
run mall headline TLPK_SOLVE_MERGE=100000000
run m256 c4 X=1
run mall c4 TLPK_SOLVE_MERGE=100000000
done
