#!/bin/bash
# Round 5, GPU session F: baseline of the restored tree (default bench line + serialised kernel stats) before the DPP potrf work.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05f
timeout 900 python bench.py --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err
S="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_prof_serial -- python bench.py $S > ${O}_prof_serial.log 2>&1
cp $(ls ${O}_prof_serial/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_serial.csv
rm -rf ${O}_prof_serial
head -c 600 ${O}_bench.json; echo; head -12 ${O}_kernel_stats_serial.csv
