#!/bin/bash
# the chain's GPU tests under the non-default strip variants: ring of two (no dynamic LDS, two workgroups per CU) with and without early entry, ring of four without early entry
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in "TLPK_CHAIN_DYNLDS=0 TLPK_CHAIN_GRID=512" "TLPK_CHAIN_DYNLDS=0 TLPK_CHAIN_GRID=512 TLPK_CHAIN_EARLY=0" "TLPK_CHAIN_EARLY=0"; do
  echo "== $v"
  env $v timeout 900 python -m pytest tests/test_chain.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
  env $v timeout 300 python bench.py --workload pds --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pds', round(d['ms_per_step'],3))"
done
