#!/bin/bash
# Round 4, GPU session N: look-ahead update of the diagonal blocks (TLPK_LOOKAHEAD) -- parity, A/B on the chain-bound and the large workloads.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -x -q 2>&1 | tail -3
B="--steps 8 --warmup 2 --no-cpu-baseline --no-host-abi --no-small-lp --no-headline --no-c3"
for rep in 1 2; do
for la in 0 1; do
  export TLPK_LOOKAHEAD=$la
  out="lookahead=$la"
  for wl in pds stair25 c4 headline; do
    r=$(timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('%.2f (potrf %.2f update %.2f reduce %.2f)' % (d['ms_per_step'], k['potrf'], k['update'], k['update_reduce']))")
    out="$out | $wl $r"
  done
  echo "$out"
done
done
for la in 0 1; do echo "lookahead=$la"; TLPK_LOOKAHEAD=$la NLIST=8 timeout 200 python tools/rank_local_timing.py 2>&1 | grep nranks; done
for la in 0 1; do echo "c3 lookahead=$la"; TLPK_LOOKAHEAD=$la timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-small-lp --no-headline --no-c3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), round(d['roofline']['frac'],4))"; done
