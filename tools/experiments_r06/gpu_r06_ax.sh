#!/bin/bash
# who gives up: the eight-shards tests in a loop with TLPK_CHAIN_DEBUG=1, output of the failing runs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
N=${N:-80}
fails=0
for i in $(seq 1 $N); do
  env $EXTRA TLPK_CHAIN_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "eight_shards" --tb=line -s > /tmp/o.txt 2>&1
  if grep -q "replayed" /tmp/o.txt; then echo "run $i: replay"; fi; if grep -q "failed" /tmp/o.txt; then fails=$((fails+1)); cp /tmp/o.txt gpurun_out/r06ax_fail_$fails.txt; grep -iE "gave up|k_chain|slot|item|role" /tmp/o.txt | head -12 | cut -c1-300; fi
done
echo "== $EXTRA: $fails failing runs of $N"
