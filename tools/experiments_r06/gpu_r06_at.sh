#!/bin/bash
# ablation (wrong numbers, valid timing): k_update's main loop without its staging -- ns1: no global loads and no LDS writes in the loop; ns2: the global loads stay, no LDS writes.
# Upper bound of what operand blocks straight to LDS (global_load_lds) could give the bulk kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06at
for rep in 1 2; do
for lib in base ns1 ns2; do
    if [ $lib = base ]; then unset TLPK_LIB; else export TLPK_LIB=$PWD/tulip.jl_amd/libtlpk_$lib.so; fi
    timeout 300 python tools/experiments_r06/update_time_tolerant.py 2>&1 | tail -1
done
done
