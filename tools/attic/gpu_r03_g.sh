#!/bin/bash
# Round 3, GPU session G: extend-add knobs (resident workgroups via extra LDS, parent columns per workgroup).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-host-abi --no-headline --unpaired"
run() { python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernel_ms']
print('$1: ms/step %.2f  extend_add %.3f  update %.2f' % (d['ms_per_step'], k['extend_add'], k['update']))"; }
{
run "default"
TLPK_EA_LDS=28672 run "EA_LDS=28672 (4 wg/CU)"
TLPK_EA_LDS=69632 run "EA_LDS=69632 (2 wg/CU)"
TLPK_EA_LDS=8192 run "EA_LDS=8192 (6-7 wg/CU)"
TLPK_EA_COLS=8 run "EA_COLS=8"
TLPK_EA_COLS=8 TLPK_EA_LDS=28672 run "EA_COLS=8 EA_LDS=28672"
TLPK_EA_COLS=4 run "EA_COLS=4"
} > gpurun_out/r03_extend_add_knobs.txt 2>&1
cat gpurun_out/r03_extend_add_knobs.txt
