#!/bin/bash
# k_update's slabs global -> LDS directly (TLPK_UPD_DMA): parity tests, then A/B against the previous build (libtlpk_ab_old.so) on C4, the north-star LP and the pds-class LP
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06au
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_chain.py tests/test_k2.py -m gpu -x -q 2>&1 | tail -3
S="--steps 8 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi --no-small-lp --no-c3"
for rep in 1 2; do
for wl in pds c4; do
for lib in new old; do
    if [ $lib = old ]; then export TLPK_LIB=$PWD/tulip.jl_amd/libtlpk_ab_old.so; else unset TLPK_LIB; fi
    timeout 600 python bench.py --workload $wl $S > ${O}_b.json 2> ${O}_b.err
    python -c "
import json; d=json.load(open('${O}_b.json')); r=d.get('roofline') or {}; print('$lib $wl', round(d['ms_per_step'],3), d.get('ms_per_step_runs'), 'k_update frac', round(r.get('frac') or 0,4), 'executed', round(r.get('frac_executed') or 0,4))"
done
done
done
