#!/bin/bash
# whole GPU suite + smoke
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_pytest_gpu.txt 2>&1
tail -15 gpurun_out/r06_pytest_gpu.txt
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2
