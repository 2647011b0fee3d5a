#!/bin/bash
# kernel trace of the default (concurrent) schedule: where does the device idle inside a Newton step?  -> gpurun_out/trace_c4_kernels.csv
# (tools/timeline_overlap.py, tools/solve_timeline.py read it)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -rf gpurun_out/trace_conc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_conc -- python bench.py --workload ${WL:-c4} --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3 > gpurun_out/trace_conc.log 2>&1
cp $(ls gpurun_out/trace_conc/*/*kernel_trace.csv | head -1) gpurun_out/trace_${WL:-c4}_kernels.csv; rm -rf gpurun_out/trace_conc
