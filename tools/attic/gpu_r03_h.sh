#!/bin/bash
# Round 3, GPU session H: k_update with 8 waves per workgroup (2 x 4 waves of 64 x 32) against the 4-wave kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TLPK_UPD_WAVES=8 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -q --maxfail=5 -k "not bench" > gpurun_out/r03_h_pytest.txt 2>&1
tail -6 gpurun_out/r03_h_pytest.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-host-abi --unpaired"
run() { python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['kernel_ms']; r=d['roofline']; h=d['headline']
print('$1: ms/step %.2f  update %.2f  frac %.3f exec %.3f | headline %.2f update %.2f frac %.3f' % (d['ms_per_step'], k['update'], r['frac'], r['frac_executed'], h['ms_per_step'], h['kernel_ms']['update'], h['roofline']['frac']))"; }
{
run "4 waves (default)"
TLPK_UPD_WAVES=8 run "8 waves"
} > gpurun_out/r03_update_8waves.txt 2>&1
cat gpurun_out/r03_update_8waves.txt
