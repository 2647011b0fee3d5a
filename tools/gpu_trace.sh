#!/bin/bash
# kernel trace of the default (concurrent) schedule: where does the device idle inside a Newton step?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -rf gpurun_out/trace_conc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_conc -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi > gpurun_out/trace_conc.log 2>&1
ls gpurun_out/trace_conc/*/ | head
