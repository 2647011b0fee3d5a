#!/bin/bash
# Round 3, GPU session F: amalgamation constants on the headline instance (padded flops vs extend-add), headline HSD parity HIP vs CPU.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--workload headline --steps 5 --warmup 2 --no-cpu-baseline --no-small-lp --no-host-abi --unpaired"
{
for gt in 400 200 100 50 0; do
  TLPK_RELAX_GAMMA_TALL=$gt timeout 300 python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=d['kernel_ms']
print('GAMMA_TALL=$gt: ms/step %.2f  frac %.3f exec %.3f  exec/alg %.3f  update %.2f extend_add %.2f trsm %.2f potrf %.2f  stored/nnzL %.2f' % (d['ms_per_step'], r['frac'], r['frac_executed'], r['flops_executed_per_step']/r['flops_per_step'], k['update'], k['extend_add'], k['trsm'], k['potrf'], d['config']['stored_over_nnzL']))"
done
for g in 10 50; do
  TLPK_RELAX_GAMMA=$g timeout 300 python bench.py $B 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=d['kernel_ms']
print('GAMMA=$g: ms/step %.2f  frac %.3f exec %.3f  exec/alg %.3f  update %.2f extend_add %.2f' % (d['ms_per_step'], r['frac'], r['frac_executed'], r['flops_executed_per_step']/r['flops_per_step'], k['update'], k['extend_add']))"
done
} > gpurun_out/r03_amalgamation_headline.txt 2>&1
cat gpurun_out/r03_amalgamation_headline.txt
HEADLINE=1 NB=100 ALGS=HSD timeout 1200 python tools/ipm_parity_at_scale.py > gpurun_out/r03_ipm_parity_headline_hsd.txt 2>&1
tail -4 gpurun_out/r03_ipm_parity_headline_hsd.txt
