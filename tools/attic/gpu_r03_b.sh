#!/bin/bash
# Round 3, GPU session B: the GPU suite (new: block detection, multi-device changes, IPM parity at scale, bench branches) + the default bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_b_pytest.txt 2>&1
tail -15 gpurun_out/r03_b_pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_b_bench.json 2> gpurun_out/r03_b_bench.err
head -c 1500 gpurun_out/r03_b_bench.json; tail -3 gpurun_out/r03_b_bench.err
