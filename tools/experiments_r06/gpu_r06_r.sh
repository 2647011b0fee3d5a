#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06r
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --workload pds $S > ${O}_bench_pds_$name.json 2> ${O}_bench_pds_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_pds_$name.json")); print("pds $name", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
    echo -n "rank-local $name: "; env "$@" NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "nranks" | cut -c90-300
}
run default X=1
run lafull TLPK_LA_FULL=1
run lafull_ks768 TLPK_LA_FULL=1 TLPK_KSPLIT_LEN=768
run lafull_ks1024 TLPK_LA_FULL=1 TLPK_KSPLIT_LEN=1024
run lafull_ks768_jit TLPK_LA_FULL=1 TLPK_KSPLIT_LEN=768 TLPK_CHAIN_JIT=1
TLPK_LA_FULL=1 TLPK_KSPLIT_LEN=768 timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds_lafull_ks768.txt 2>&1
head -6 ${O}_chain_trace_pds_lafull_ks768.txt | cut -c1-250
