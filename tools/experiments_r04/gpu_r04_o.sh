#!/bin/bash
# Round 4, GPU session O: zero-fill + assembly of the upper fronts beside the leaf levels (TLPK_DEFER_UPPER) -- parity, A/B.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_hsd_device.py -m gpu -x -q 2>&1 | tail -3
B="--steps 10 --warmup 3 --no-cpu-baseline --no-host-abi --no-small-lp --no-headline --no-c3 --no-roofline"
for rep in 1 2 3; do
for v in 0 1; do
  export TLPK_DEFER_UPPER=$v
  out="defer_upper=$v"
  for wl in c4 headline; do
    r=$(timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f (unpaired %.2f)' % (d['ms_per_step'], d.get('unpaired_ms_per_step', 0)))")
    out="$out | $wl $r"
  done
  echo "$out"
done
done
