#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
try() { local name=$1; shift; local f=0
  for i in $(seq 1 40); do
    env "$@" timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eight_shards_enqueue" 2>&1 | grep -q "failed" && f=$((f+1))
  done; echo "$name: $f of 40 runs failed"; }
try serial_default X=1
try serial_hwq16 GPU_MAX_HW_QUEUES=16
