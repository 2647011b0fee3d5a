#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06v
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { local name=$1; shift
    env "$@" timeout 300 python bench.py --workload pds $S > ${O}_bench_pds_$name.json 2> ${O}_bench_pds_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_pds_$name.json")); print("pds $name", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
    echo -n "rank-local $name: "; env "$@" NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "nranks" | cut -c90-300
}
run default X=1
run min513 TLPK_CHAIN_MIN_NS=513
run min257 TLPK_CHAIN_MIN_NS=257
run twoper TLPK_CHAIN_DYNLDS=0 TLPK_CHAIN_GRID=512
run grid512 TLPK_CHAIN_GRID=512
