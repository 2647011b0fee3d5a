#!/bin/bash
# Round 4, GPU session B: per-kernel time of a Newton step (serialised schedule) for C4 and the north-star instance.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
S="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp"
for wl in c4 headline; do
  TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04b_prof_$wl -- python bench.py --workload $wl $S > gpurun_out/r04b_prof_$wl.log 2>&1
  cp $(ls gpurun_out/r04b_prof_$wl/*/*kernel_stats.csv | head -1) gpurun_out/r04b_kernel_stats_serial_$wl.csv
  rm -rf gpurun_out/r04b_prof_$wl
  echo "== $wl"; cut -d, -f1-7 gpurun_out/r04b_kernel_stats_serial_$wl.csv | sed 's/tlpk:://; s/(.*)//' | head -32
done
