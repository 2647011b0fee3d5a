#!/bin/bash
# Round 5, GPU session R: one stream group against two for the rank-local work at N = 2, 4, 8 (32 / 16 / 8 diagonal blocks per rank), C4 and north-star shape
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05r
for n in 8 4 2; do for ng in 1 2; do echo "c4 N=$n NG=$ng: $(NG=$ng NLIST=$n timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1 | cut -c1-300)"; done; done | tee ${O}_groups.txt
for n in 8 4; do for ng in 1 2; do echo "north-star N=$n NG=$ng: $(HEADLINE=1 NG=$ng NLIST=$n timeout 400 python tools/rank_local_timing.py 2>&1 | tail -1 | cut -c1-300)"; done; done | tee -a ${O}_groups.txt
