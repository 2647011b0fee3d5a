#!/bin/bash
# Round 3, GPU session K: balanced block mapping of diagonal / column-limited k_update tiles: parity tests, then A/B of the step time.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_lp_configs.py -m gpu -q --maxfail=6 > gpurun_out/r03_k_pytest.txt 2>&1
tail -5 gpurun_out/r03_k_pytest.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-small-lp --no-host-abi"
run() { TLPK_UPD_REMAP=$2 python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); h=d.get('headline', {}); r=d['roofline']; hr=h.get('roofline', {})
print('$1: ms/step %.3f | k_update %.3f ms frac %.4f exec %.4f | headline %.3f ms/step frac %.4f' % (d['ms_per_step'], d['kernel_ms']['update'], r['frac'], r['frac_executed'], h.get('ms_per_step', 0), hr.get('frac', 0)))"; }
{
run "ordinary mapping (TLPK_UPD_REMAP=14)" 14
run "balanced diagonal + column-limited tiles (default)" 2
run "balanced diagonal tiles only (TLPK_UPD_REMAP=10)" 10
run "ordinary mapping (TLPK_UPD_REMAP=14)" 14
run "balanced diagonal + column-limited tiles (default)" 2
} > gpurun_out/r03_update_diag_tiles.txt 2>&1
cat gpurun_out/r03_update_diag_tiles.txt
