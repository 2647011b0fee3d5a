#!/bin/bash
# polling policy of the early strips (TLPK_EARLY_QUIET): new library against libtlpk_ab_old.so (polls every 0.2 - 0.8 us throughout)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06as
timeout 900 python -m pytest tests/test_chain.py -m gpu -x -q 2>&1 | tail -2
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
for rep in 1 2 3; do
for lib in new old; do
    if [ $lib = old ]; then export TLPK_LIB=$PWD/tulip.jl_amd/libtlpk_ab_old.so; else unset TLPK_LIB; fi
    timeout 300 python bench.py --workload pds $S > ${O}_b.json 2> ${O}_b.err
    python -c "
import json; d=json.load(open('${O}_b.json')); print('$lib pds', round(d['ms_per_step'],3), d.get('ms_per_step_runs'))"
done
done
for lib in new old; do
    if [ $lib = old ]; then export TLPK_LIB=$PWD/tulip.jl_amd/libtlpk_ab_old.so; else unset TLPK_LIB; fi
    NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "nranks" | cut -c1-140
    timeout 300 python tools/chain_trace.py > ${O}_trace_$lib.txt 2>&1
    sed -n 2,5p ${O}_trace_$lib.txt | cut -c1-200
    sed -n 8,12p ${O}_trace_$lib.txt | cut -c1-230; tail -3 ${O}_trace_$lib.txt | cut -c1-230
done
