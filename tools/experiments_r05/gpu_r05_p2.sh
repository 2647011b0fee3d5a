#!/bin/bash
# Round 5, session P2: the kernel-agreement test alone
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "diagonal_block_kernels_agree" 2>&1 | grep "max |L3\|passed\|failed\|Error\|assert" | cut -c1-300 | tee gpurun_out/r05p2_agree.txt
