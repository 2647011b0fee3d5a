#!/bin/bash
# Round 5, session AA (host only, on the GPU box's CPU quota): phases of the host analyse of the two bench shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TLPK_TIMING=1 timeout 300 python tools/analyse_phases.py 2>&1 | tail -42 > gpurun_out/r05_analyse_phases_c4.txt
HEADLINE=1 TLPK_TIMING=1 timeout 300 python tools/analyse_phases.py 2>&1 | tail -42 > gpurun_out/r05_analyse_phases_headline.txt
tail -40 gpurun_out/r05_analyse_phases_c4.txt; tail -40 gpurun_out/r05_analyse_phases_headline.txt
