#!/bin/bash
# Round 3, GPU session Q: staggered extend-add launches of the two stream groups (TLPK_STAGGER), A/B on C4 and the north-star instance.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 20 --warmup 5 --no-cpu-baseline --no-small-lp --no-host-abi --no-roofline"
run() { TLPK_STAGGER=$2 python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); h=d.get('headline', {})
print('$1: ms/step %.3f (unpaired %.3f) | headline %.3f' % (d['ms_per_step'], d.get('unpaired_ms_per_step', 0), h.get('ms_per_step', 0)))"; }
{
run "lockstep groups (default)" 0
run "staggered extend-add (launches >= 10000 tasks)" 1
run "staggered extend-add (launches >= 1000 tasks)" 1000
run "lockstep groups (default)" 0
run "staggered extend-add (launches >= 10000 tasks)" 1
} > gpurun_out/r03_stagger.txt 2>&1
cat gpurun_out/r03_stagger.txt
