#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06k
timeout 600 python -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -15
