#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for S in 2 3 4; do
  for Q in 8 16; do
  GPU_MAX_HW_QUEUES=$Q TLPK_STREAMS=$S timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams',$S,'queues',$Q,'ms/step',round(d['ms_per_step'],2), d['config']['stream_groups'])"
  done
done
