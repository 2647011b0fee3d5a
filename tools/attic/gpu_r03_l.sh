#!/bin/bash
# Round 3, GPU session L: wave -> SIMD placement probe, refinement on sharded / multi-device handles, A/B of the diagonal-tile mapping
# under the other placement assumption.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
tools/wave_simd_map > gpurun_out/r03_wave_simd_map.txt 2>&1; cat gpurun_out/r03_wave_simd_map.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=6 -k "refinement or sharded_split_phase or multi_device" > gpurun_out/r03_l_pytest.txt 2>&1
tail -12 gpurun_out/r03_l_pytest.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-small-lp --no-host-abi --no-headline"
run() { TLPK_UPD_REMAP=$2 python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']
print('$1: ms/step %.3f | k_update %.3f ms frac %.4f exec %.4f' % (d['ms_per_step'], d['kernel_ms']['update'], r['frac'], r['frac_executed']))"; }
{
run "ordinary mapping (TLPK_UPD_REMAP=14)" 14
run "balanced, waves w / w+4 share a SIMD (2)" 2
run "balanced, waves 2s / 2s+1 share a SIMD (18)" 18
run "ordinary mapping (TLPK_UPD_REMAP=14)" 14
run "balanced, waves 2s / 2s+1 share a SIMD (18)" 18
} > gpurun_out/r03_update_diag_tiles_b.txt 2>&1
cat gpurun_out/r03_update_diag_tiles_b.txt
