#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for kw in 256 2048; do timeout 60 tools/update_bench_v0t 64 4525 3530 $kw 1 | tail -1; done
for kw in 256 512 2048; do timeout 60 tools/update_bench_v0 64 4525 3530 $kw 1 | tail -1; done
timeout 60 tools/update_bench_v0 64 4525 3530 256 0 | tail -1
bash tools/gpu_quick.sh
