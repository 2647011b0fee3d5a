#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06t
WL=c4 BLOCKS=8 timeout 300 python tools/chain_trace.py > ${O}_chain_trace_c4_8blocks.txt 2>&1
grep -v "^  k0=" ${O}_chain_trace_c4_8blocks.txt | cut -c1-250 | head -60
NLIST=8 timeout 300 python tools/rank_local_timing.py 2>&1 | tail -30 | cut -c1-300
