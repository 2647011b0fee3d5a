#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06q
timeout 900 python -m pytest tests/test_chain.py tests/test_gpu_parity.py tests/test_k2.py -m gpu -x -q 2>&1 | tail -3
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
run ks0 TLPK_KSPLIT_LEN=0
run ks512 TLPK_KSPLIT_LEN=512
run ks1024 TLPK_KSPLIT_LEN=1024
run jit0 TLPK_CHAIN_JIT=0
run jit0ks0 TLPK_CHAIN_JIT=0 TLPK_KSPLIT_LEN=0
timeout 300 python bench.py --workload c4 $S > ${O}_bench_c4.json 2> ${O}_bench_c4.err
python - <<P
import json
d=json.load(open("${O}_bench_c4.json")); print("c4", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds.txt 2>&1
head -6 ${O}_chain_trace_pds.txt | cut -c1-250
