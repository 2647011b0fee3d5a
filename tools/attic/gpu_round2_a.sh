#!/bin/bash
# GPU call A of round 2: parity suite, baseline bench with the new measurement fields, MFMA / HBM counters.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
nproc > gpurun_out/r2a_host.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r2a_host.txt; free -g | head -2 >> gpurun_out/r2a_host.txt
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -150 > gpurun_out/r2a_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
# MFMA utilisation on the panel GEMMs (own run, kernel-trace only)
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r2a_pmc_mfma -- \
    python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi > gpurun_out/r2a_pmc_mfma.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r2a_pmc_$c -- \
        python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi > gpurun_out/r2a_pmc_$c.log 2>&1
done
python tools/pmc_summarise.py gpurun_out/r2a_pmc_mfma gpurun_out/r2a_pmc_FETCH_SIZE gpurun_out/r2a_pmc_WRITE_SIZE > gpurun_out/r2a_pmc_summary.md 2>&1
tail -5 gpurun_out/r2a_tests.log
