#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06o
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eight_shards_enqueue" 2>&1 | tail -30 | cut -c1-250
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
for mt in 256 512 1024 2048; do
    TLPK_MACRO_TILES=$mt timeout 300 python bench.py --workload pds $S > ${O}_bench_pds_mt$mt.json 2> ${O}_bench_pds_mt$mt.err
    python - <<P
import json
d=json.load(open("${O}_bench_pds_mt$mt.json")); print("pds MACRO_TILES=$mt", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
done
for mt in 256 512 1024; do
echo "== rank-local MACRO_TILES=$mt"
TLPK_MACRO_TILES=$mt NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "nranks" | cut -c1-300
done
TLPK_MACRO_TILES=512 timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds_mt512.txt 2>&1
head -6 ${O}_chain_trace_pds_mt512.txt | cut -c1-250
