#!/bin/bash
# GPU call F: single-process multi-device mode + full regression
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "multi_device" 2>&1 | tail -30 > gpurun_out/r2f_multi.log; tail -12 gpurun_out/r2f_multi.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r2f_tests.log; tail -3 gpurun_out/r2f_tests.log
