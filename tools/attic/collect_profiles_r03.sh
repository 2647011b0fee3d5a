#!/bin/bash
# Round-3 measurement artifacts, collected on the GPU box (run through gpurun from the repo root); everything lands in
# gpurun_out/r03_* and is then copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
nproc > gpurun_out/r03_host.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r03_host.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err
B="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --unpaired"
S="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp"
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_prof_serial -- python bench.py $S > gpurun_out/r03_prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_prof_concurrent -- python bench.py $S > gpurun_out/r03_prof_concurrent.log 2>&1
for wl in c4 headline; do
  for c in FETCH_SIZE WRITE_SIZE; do
    TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r03_pmc_${wl}_$c -- python bench.py --workload $wl $B > gpurun_out/r03_pmc_${wl}_$c.log 2>&1
  done
  python tools/pmc_to_json.py $wl gpurun_out/r03_pmc_${wl}_FETCH_SIZE gpurun_out/r03_pmc_${wl}_WRITE_SIZE gpurun_out/r03_pmc_k_update.json \
    "round 3: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes), TLPK_STREAMS=1 TLPK_SERIAL=1 python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --unpaired (tools/collect_profiles_r03.sh); FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md section HBM; calibration profiles/r01_pmc_k_update_hbm_traffic.md); NOT collected in the bench run itself" >> gpurun_out/r03_pmc_to_json.log 2>&1
done
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r03_pmc_mfma -- python bench.py $B > gpurun_out/r03_pmc_mfma.log 2>&1
python tools/pmc_summarise.py gpurun_out/r03_pmc_c4_FETCH_SIZE gpurun_out/r03_pmc_c4_WRITE_SIZE gpurun_out/r03_pmc_mfma > gpurun_out/r03_pmc_summary.md 2>&1
python tools/pmc_summarise.py gpurun_out/r03_pmc_headline_FETCH_SIZE gpurun_out/r03_pmc_headline_WRITE_SIZE > gpurun_out/r03_pmc_summary_headline.md 2>&1
cp $(ls gpurun_out/r03_prof_serial/*/*kernel_stats.csv | head -1) gpurun_out/r03_final_kernel_stats_serial.csv
cp $(ls gpurun_out/r03_prof_concurrent/*/*kernel_stats.csv | head -1) gpurun_out/r03_final_kernel_stats_concurrent.csv
timeout 600 python bench.py --workload c3 --c3-rows 50000 --steps 3 --warmup 1 --no-host-abi > gpurun_out/r03_c3_50k_rows_bench.json 2> gpurun_out/r03_c3.err
NLIST=1,2,4,8 timeout 600 python tools/rank_local_timing.py > gpurun_out/r03_rank_local_timing.txt 2>&1
timeout 300 python tools/solve_c4_lp.py > gpurun_out/r03_c4_lp_end_to_end.txt 2>&1
HEADLINE=1 timeout 400 python tools/solve_c4_lp.py > gpurun_out/r03_headline_lp_end_to_end.txt 2>&1
NSHARDS=2 timeout 400 python tools/solve_c4_lp.py > gpurun_out/r03_c4_lp_two_shards_one_gpu.txt 2>&1
rm -rf gpurun_out/r03_prof_serial gpurun_out/r03_prof_concurrent gpurun_out/r03_pmc_c4_* gpurun_out/r03_pmc_headline_* gpurun_out/r03_pmc_mfma
tail -3 gpurun_out/r03_rank_local_timing.txt; cat gpurun_out/r03_pmc_to_json.log; head -c 400 gpurun_out/r03_final_bench.json
