#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT/_old"
f=0
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eight_shards_enqueue" 2>&1 | grep -q "failed" && f=$((f+1))
done; echo "pre-session library (a322b6d): $f of 12 runs failed"
