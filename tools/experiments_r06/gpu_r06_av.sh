#!/bin/bash
# ablations of k_update's main loop behind item 13 (wrong numbers, valid timing; tools/experiments_r06/update_time_tolerant.py): a1 = no closing barrier / wait per round, a3 = no LDS operand reads
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for lib in base a2; do
    if [ $lib = base ]; then unset TLPK_LIB; else export TLPK_LIB=$PWD/tulip.jl_amd/libtlpk_$lib.so; fi
    timeout 300 python tools/experiments_r06/update_time_tolerant.py 2>&1 | tail -1
done
done
