#!/bin/bash
# Round-2 measurement artifacts, collected on the GPU box (run through gpurun from the repo root); everything lands in
# gpurun_out/r02_* and is then copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
nproc > gpurun_out/r02_host.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r02_host.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi"
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof_serial -- \
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi > gpurun_out/r02_prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof_concurrent -- \
    python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi > gpurun_out/r02_prof_concurrent.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r02_pmc_$c -- $B > gpurun_out/r02_pmc_$c.log 2>&1
done
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r02_pmc_mfma -- $B > gpurun_out/r02_pmc_mfma.log 2>&1
python tools/pmc_summarise.py gpurun_out/r02_pmc_FETCH_SIZE gpurun_out/r02_pmc_WRITE_SIZE gpurun_out/r02_pmc_mfma > gpurun_out/r02_pmc_summary.md 2>&1
cp $(ls gpurun_out/r02_prof_serial/*/*kernel_stats.csv | head -1) gpurun_out/r02_final_kernel_stats_serial.csv
cp $(ls gpurun_out/r02_prof_concurrent/*/*kernel_stats.csv | head -1) gpurun_out/r02_final_kernel_stats_concurrent.csv
timeout 900 python bench.py --workload c3 --c3-rows 50000 --steps 3 --warmup 1 --no-host-abi > gpurun_out/r02_c3_50k_rows_bench.json 2> gpurun_out/r02_c3.err
NLIST=1,2,4,8 timeout 900 python tools/rank_local_timing.py > gpurun_out/r02_rank_local_timing.txt 2>&1
rm -rf gpurun_out/r02_prof_serial gpurun_out/r02_prof_concurrent
tail -3 gpurun_out/r02_rank_local_timing.txt; head -c 600 gpurun_out/r02_final_bench.json
