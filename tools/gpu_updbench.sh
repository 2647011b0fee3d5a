#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|Counter)?.*SQ_(WAIT|LDS|BUSY|WAVE|VALU_MFMA|INSTS_VALU_MFMA|ACTIVE|INST_CYCLES|LEVEL)" | head -60 > gpurun_out/avail.txt
wc -l gpurun_out/avail.txt
for cset in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS"; do
  tag=$(echo $cset | cut -d' ' -f1)
  for mode in 1 0; do
    timeout 120 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d gpurun_out/ub_${tag}_$mode -- tools/update_bench_v0 64 4525 3530 2048 $mode > gpurun_out/ub_${tag}_$mode.log 2>&1
    echo "== $cset mode(left=1) $mode"; python tools/pmc_summarise.py gpurun_out/ub_${tag}_$mode | grep -E "k_update"
  done
done
