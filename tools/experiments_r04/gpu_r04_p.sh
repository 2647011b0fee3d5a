#!/bin/bash
# Round 4, GPU session P: the upper fronts' zero-fill on a CU-masked stream (TLPK_ZCUS CUs), so that it leaves HBM bandwidth to the leaf levels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 10 --warmup 3 --no-cpu-baseline --no-host-abi --no-small-lp --no-headline --no-c3 --no-roofline"
for rep in 1 2; do
for v in off 0 32 64 128; do
  if [ $v = off ]; then export TLPK_DEFER_UPPER=0; unset TLPK_ZCUS; else export TLPK_DEFER_UPPER=1 TLPK_ZCUS=$v; fi
  out="zcus=$v"
  for wl in c4 headline; do
    r=$(timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % d['ms_per_step'])")
    out="$out | $wl $r"
  done
  echo "$out"
done
done
