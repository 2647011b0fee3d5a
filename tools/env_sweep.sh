#!/bin/bash
# A/B runs of the default bench under different environments (run on the GPU box via gpurun):
#   bash tools/env_sweep.sh "TLPK_STREAMS=1" "TLPK_STREAMS=2" "TLPK_STREAMS=2 GPU_MAX_HW_QUEUES=4"
# Prints ms/step and the per-class kernel times of each configuration.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
    echo "== $cfg"
    env $cfg timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'], 2), 'solve', round(d['solve_roofline']['ms_per_solve'],3), 'k_update TFLOP/s', round(d['roofline']['achieved'], 1), 'nnzL_stored', d['config']['nnzL_stored'], 'nsup', d['config']['n_supernodes'])
print(d['kernel_ms'])"
done
