#!/bin/bash
# GPU call G: what keeps the matrix cores idle 28 % of k_update?  SQ counters, two passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi"
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/r2g_pmc1 -- $B > gpurun_out/r2g_pmc1.log 2>&1
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_IFETCH --output-format csv -d gpurun_out/r2g_pmc2 -- $B > gpurun_out/r2g_pmc2.log 2>&1
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/r2g_pmc3 -- $B > gpurun_out/r2g_pmc3.log 2>&1
python tools/pmc_summarise.py gpurun_out/r2g_pmc1 gpurun_out/r2g_pmc2 gpurun_out/r2g_pmc3 2>&1 | grep -E "k_update \||k_trsm \||k_extend_add|kernel \||---"
