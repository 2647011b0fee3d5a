#!/bin/bash
# Round 5, GPU session H: potrf_block_dpp as the default -- phase stamps of one block (trace build), then the whole GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05h
for a in "1 64" "1 256"; do timeout 60 tools/potrf_wave_bench_trace $a; done 2>&1 | tee ${O}_potrf_trace.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee ${O}_pytest.txt
