#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python tools/chain_trace.py > gpurun_out/r06b_chain_trace_pds.txt 2>&1
head -70 gpurun_out/r06b_chain_trace_pds.txt
