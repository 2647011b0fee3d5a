#!/bin/bash
# Round 4, GPU session H: rank-local work at N = 1, 2, 4, 8 (C4 and the north-star LP) for the scaling projection.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
NLIST=1,2,4,8 timeout 600 python tools/rank_local_timing.py > gpurun_out/r04_rank_local_c4.txt 2>&1
HEADLINE=1 NLIST=1,2,4,8 timeout 900 python tools/rank_local_timing.py > gpurun_out/r04_rank_local_headline.txt 2>&1
grep nranks gpurun_out/r04_rank_local_c4.txt gpurun_out/r04_rank_local_headline.txt | cut -c1-330
python tools/scale_projection.py gpurun_out/r04_rank_local_c4.txt gpurun_out/r04_rank_local_headline.txt | tee gpurun_out/r04_scale_projection.txt
