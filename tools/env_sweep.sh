#!/bin/bash
# A/B runs of the default bench under different environments (run on the GPU box via gpurun):
#   bash tools/env_sweep.sh "TLPK_STREAMS=1" "TLPK_STREAMS=2" "TLPK_STREAMS=2 GPU_MAX_HW_QUEUES=4"
# Prints ms/step and the per-class kernel times of each configuration.
for cfg in "$@"; do
    echo "== $cfg"
    env $cfg timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms/step', round(d['ms_per_step'], 2), 'k_update TFLOP/s', round(d['roofline']['achieved'], 1))
print(d['kernel_ms'])"
done
