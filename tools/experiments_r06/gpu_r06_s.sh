#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06s
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { local name=$1; shift
    env "$@" timeout 300 python bench.py --workload c4 $S > ${O}_bench_c4_$name.json 2> ${O}_bench_c4_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_c4_$name.json")); print("c4 $name", round(d["ms_per_step"],3), d.get("ms_per_step_runs"), "groups", d["config"].get("stream_groups"))
P
}
run default X=1
run streams3 TLPK_STREAMS=3
run streams4 TLPK_STREAMS=4
run lpt TLPK_UPD_LPT=1
run chain1 TLPK_CHAIN=1
run chainmax16 TLPK_CHAIN_MAX_FRONTS=16
bash tools/gpu_trace.sh
python tools/timeline_overlap.py gpurun_out/trace_c4_kernels.csv > ${O}_timeline_c4.txt 2>&1
STEP=3 python tools/solve_timeline.py gpurun_out/trace_c4_kernels.csv > ${O}_solve_timeline_c4.txt 2>&1
rm -f gpurun_out/trace_c4_kernels.csv
tail -30 ${O}_timeline_c4.txt | cut -c1-220
