#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
WL=headline bash tools/gpu_trace.sh
STEP=3 python tools/timeline_overlap.py gpurun_out/trace_headline_kernels.csv > gpurun_out/r06ab_timeline_headline.txt 2>&1
STEP=3 python tools/solve_timeline.py gpurun_out/trace_headline_kernels.csv > gpurun_out/r06ab_solve_timeline_headline.txt 2>&1
rm -f gpurun_out/trace_headline_kernels.csv
tail -32 gpurun_out/r06ab_timeline_headline.txt | cut -c1-200; tail -9 gpurun_out/r06ab_solve_timeline_headline.txt
