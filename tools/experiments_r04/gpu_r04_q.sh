#!/bin/bash
# Round 4, GPU session Q: window of the skip decision (TLPK_SKIP_WIN) -- do tiles of a super-tile that skip the same slabs win back the L2 reuse?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--steps 8 --warmup 2 --no-cpu-baseline --no-host-abi --no-small-lp --no-headline --no-c3"
for rep in 1 2; do
for v in 128 256 512; do
  export TLPK_SKIP_WIN=$v
  out="skip_win=$v"
  for wl in c4 headline; do
    r=$(timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%.2f (update %.2f frac %.4f exec %.4f ratio %.3f)' % (d['ms_per_step'], d['kernel_ms']['update'], r['frac'], r['frac_executed'], r['flops_executed_per_step']/r['flops_per_step']))")
    out="$out | $wl $r"
  done
  echo "$out"
done
done
