#!/bin/bash
# the eight-shards-on-one-device tests in a loop: default against TLPK_CHAIN_EARLY=0 (N runs of the three tests each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${N:-25}
for v in ${VARIANTS:-"X=1 TLPK_CHAIN_EARLY=0"}; do
  fails=0
  for i in $(seq 1 $N); do
    env $v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "eight_shards" --tb=line > /tmp/o.txt 2>&1
    if grep -q "failed" /tmp/o.txt; then fails=$((fails+1)); grep -E "^/.*Error|FAILED" /tmp/o.txt | head -3 | cut -c1-250; fi
  done
  echo "== $v: $fails failing runs of $N"
done
