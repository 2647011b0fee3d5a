#!/bin/bash
# rank-local N = 8 (8 blocks of C4 on one GPU): launches vs chain (two / one workgroup per CU)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06g
run() { name=$1; shift; echo "== $name"; env "$@" timeout 300 python tools/rank_local_timing.py 2>&1 | grep -E "ms/step|nranks" ; }
{
run launch NLIST=8 TLPK_CHAIN=0
run chain_default NLIST=8 TLPK_X=1
run chain_2wg NLIST=8 TLPK_CHAIN_DYNLDS=0
run chain_2wg_1024 NLIST=8 TLPK_CHAIN_DYNLDS=0 TLPK_CHAIN_GRID=1024
run launch_again NLIST=8 TLPK_CHAIN=0
run launch_n4 NLIST=4 TLPK_CHAIN=0
run chain_n4 NLIST=4 TLPK_X=1
run chain_n4_2wg NLIST=4 TLPK_CHAIN_DYNLDS=0
} > ${O}_rank_local_variants.txt 2>&1
cat ${O}_rank_local_variants.txt | cut -c1-330
