#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06p
timeout 900 python -m pytest tests/test_chain.py tests/test_gpu_parity.py tests/test_k2.py -m gpu -x -q 2>&1 | tail -3
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
for ks in 512 768 1024 0; do
    TLPK_KSPLIT_LEN=$ks timeout 300 python bench.py --workload pds $S > ${O}_bench_pds_ks$ks.json 2> ${O}_bench_pds_ks$ks.err
    python - <<P
import json
d=json.load(open("${O}_bench_pds_ks$ks.json")); print("pds KSPLIT_LEN=$ks", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
done
TLPK_LA_MACRO=1 timeout 300 python bench.py --workload pds $S > ${O}_bench_pds_macro.json 2> ${O}_bench_pds_macro.err
python - <<P
import json
d=json.load(open("${O}_bench_pds_macro.json")); print("pds LA_MACRO=1", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
timeout 300 python bench.py --workload c4 $S > ${O}_bench_c4.json 2> ${O}_bench_c4.err
python - <<P
import json
d=json.load(open("${O}_bench_c4.json")); print("c4", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
for ks in 512 768 1024 0; do
echo "== rank-local KSPLIT_LEN=$ks"
TLPK_KSPLIT_LEN=$ks NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "nranks" | cut -c1-300
done
timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds.txt 2>&1
head -6 ${O}_chain_trace_pds.txt | cut -c1-250
