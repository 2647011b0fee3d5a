#!/bin/bash
# Round-6 measurement artifacts, collected on the GPU box (gpurun from the repo root); everything lands in gpurun_out/r06_final_* and is copied into profiles/ by hand.
# (No pytest run in front of it: after minutes of sustained load the latency-bound launches of this pool's boxes run 35 - 45 % slower, profiles/r05_potrf_dpp.txt.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06_final
nproc > ${O}_host.txt; grep -m1 "model name" /proc/cpuinfo >> ${O}_host.txt
timeout 900 python bench.py --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err
S="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_prof_serial -- python bench.py $S > ${O}_prof_serial.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_prof_concurrent -- python bench.py $S > ${O}_prof_concurrent.log 2>&1
STEP=4 python tools/timeline_overlap.py $(ls ${O}_prof_concurrent/*/*kernel_trace.csv | head -1) > ${O}_timeline_c4.txt 2>&1
STEP=4 python tools/solve_timeline.py $(ls ${O}_prof_concurrent/*/*kernel_trace.csv | head -1) > ${O}_solve_timeline_c4.txt 2>&1
cp $(ls ${O}_prof_serial/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_serial.csv
cp $(ls ${O}_prof_concurrent/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_concurrent.csv
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_prof_pds -- python bench.py --workload pds $S > ${O}_prof_pds.log 2>&1
cp $(ls ${O}_prof_pds/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_pds.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_prof_pds_conc -- python bench.py --workload pds $S > ${O}_prof_pds_conc.log 2>&1
cp $(ls ${O}_prof_pds_conc/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_pds_default.csv
rm -rf ${O}_prof_serial ${O}_prof_concurrent ${O}_prof_pds ${O}_prof_pds_conc
NLIST=1,2,4,8 timeout 600 python tools/rank_local_timing.py > ${O}_rank_local_c4.txt 2>&1
HEADLINE=1 NLIST=1,2,4,8 timeout 900 python tools/rank_local_timing.py > ${O}_rank_local_headline.txt 2>&1
python tools/scale_projection.py ${O}_rank_local_c4.txt ${O}_rank_local_headline.txt > ${O}_scale_projection.txt 2>&1
timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds.txt 2>&1
WL=c4 BLOCKS=8 timeout 300 python tools/chain_trace.py > ${O}_chain_trace_c4_8blocks.txt 2>&1
for n in 256 200 64; do timeout 120 ./tools/potrf_wave_bench 64 $n; timeout 120 ./tools/potrf_wave_bench 1 $n; done > ${O}_potrf_bench.txt 2>&1
timeout 300 python tools/solve_c4_lp.py > ${O}_c4_lp_end_to_end.txt 2>&1
head -c 300 ${O}_bench.json; echo; cat ${O}_scale_projection.txt
