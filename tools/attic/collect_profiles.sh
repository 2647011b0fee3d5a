#!/bin/bash
# Collects the round's measurement artifacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/final_bench.json         default bench line (C4)
#   gpurun_out/prof_final/              rocprofv3 --kernel-trace --stats, single stream group, TLPK_SERIAL=1 (every launch on one stream, as in the bench roofline leg)
#   gpurun_out/pmc_fetch, pmc_write     separate PMC passes (FETCH_SIZE / WRITE_SIZE), one Newton step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- \
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof_final.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/pmc_$(echo $c | tr A-Z a-z | sed 's/_size//')
    TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- \
        python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $d.log 2>&1
done
ls -R gpurun_out/prof_final gpurun_out/pmc_fetch gpurun_out/pmc_write | head -30
timeout 900 python bench.py --workload headline --steps 3 --warmup 1 > gpurun_out/headline_bench.json 2> gpurun_out/headline_bench.err
timeout 900 python bench.py --workload c3 --c3-rows 50000 --steps 2 --warmup 1 > gpurun_out/bench_c3_50k.json 2> gpurun_out/c3.err
NLIST=1,2,4,8 timeout 600 python tools/rank_local_timing.py > gpurun_out/rank_local_timing.txt 2>&1
