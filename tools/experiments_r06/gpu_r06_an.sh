#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06_final
S="--steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_prof_headline -- python bench.py --workload headline $S > ${O}_prof_headline.log 2>&1
cp $(ls ${O}_prof_headline/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_headline_serial.csv
rm -rf ${O}_prof_headline
head -8 ${O}_kernel_stats_headline_serial.csv | cut -c1-150
