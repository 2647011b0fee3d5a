#!/bin/bash
# Round 3, GPU session C: whole GPU suite + default bench line (paired solves, headline CPU baseline, small-LP legs).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r03_c_pytest.txt 2>&1
tail -40 gpurun_out/r03_c_pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_c_bench.json 2> gpurun_out/r03_c_bench.err
head -c 700 gpurun_out/r03_c_bench.json; tail -3 gpurun_out/r03_c_bench.err
