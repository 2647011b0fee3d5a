#!/bin/bash
# Round 5, GPU session L: potrf_block_dpp (row-oriented, one trsm call per step) as the default -- whole GPU suite, default bench line, serialised kernel stats.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05l
for a in "1 64" "64 64" "64 256" "64 200" "64 37"; do timeout 60 tools/potrf_wave_bench $a | head -1; done 2>&1 | tee ${O}_potrf_bench.txt
for a in "1 64" "1 256"; do timeout 60 tools/potrf_wave_bench_trace $a | tail -1; done 2>&1 | tee -a ${O}_potrf_bench.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee ${O}_pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err
S="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_prof_serial -- python bench.py $S > ${O}_prof_serial.log 2>&1
cp $(ls ${O}_prof_serial/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_serial.csv
rm -rf ${O}_prof_serial
head -c 400 ${O}_bench.json; echo; head -8 ${O}_kernel_stats_serial.csv
