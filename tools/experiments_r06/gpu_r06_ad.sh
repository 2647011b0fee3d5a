#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06ad
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { local name=$1; shift
    env "$@" timeout 300 python bench.py --workload c4 $S > ${O}_bench_c4_$name.json 2> ${O}_bench_c4_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_c4_$name.json")); print("c4 $name", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
}
run default X=1
run stagger TLPK_STAGGER=1
run super2 TLPK_UPD_SUPER=2
run super8 TLPK_UPD_SUPER=8
run remap1 TLPK_UPD_REMAP=1
run hwq4 GPU_MAX_HW_QUEUES=4
run eacols8 TLPK_EA_COLS=8
run default2 X=1
