#!/bin/bash
# Round 3, GPU session I: blocked 64 x 64 diagonal-block factorisation (potrf_block_fast) -- parity suite, then old vs new on the chain-bound LPs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_abi.py tests/test_blocks.py -m gpu -q --maxfail=6 > gpurun_out/r03_i_pytest.txt 2>&1
tail -8 gpurun_out/r03_i_pytest.txt
B="--steps 20 --warmup 3 --no-cpu-baseline --no-small-lp --no-host-abi --no-headline --unpaired"
run() { python bench.py $B $2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d.get('kernel_ms', {})
print('$1: ms/step %.3f  potrf %.3f trsm %.3f update %.3f  residuals %s' % (d['ms_per_step'], k.get('potrf', 0), k.get('trsm', 0), k.get('update', 0), d['config']['residual_inf']))"; }
{
TLPK_POTRF=old run "pds   potrf=old" "--workload pds"
run "pds   potrf=blocked" "--workload pds"
TLPK_POTRF=old run "stair potrf=old" "--workload stair25"
run "stair potrf=blocked" "--workload stair25"
TLPK_POTRF=old run "c4    potrf=old" ""
run "c4    potrf=blocked" ""
TLPK_POTRF=old NLIST=8 python tools/rank_local_timing.py 2>/dev/null | tail -2
NLIST=8 python tools/rank_local_timing.py 2>/dev/null | tail -2
} > gpurun_out/r03_potrf_blocked.txt 2>&1
cat gpurun_out/r03_potrf_blocked.txt
