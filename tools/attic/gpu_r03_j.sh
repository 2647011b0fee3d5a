#!/bin/bash
# Round 3, GPU session J: asynchronous update (root front on its own stream): tests, then blocking vs asynchronous step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=6 -k "asynchronous or fused_factor or paired_h or device_hsd or hsd or mpc or model or bench_line" > gpurun_out/r03_j_pytest.txt 2>&1
tail -8 gpurun_out/r03_j_pytest.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-small-lp --no-host-abi --no-roofline"
run() { python bench.py $B $2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); h=d.get('headline', {})
print('$1: ms/step %.3f (unpaired+blocking %.3f) | headline %.3f' % (d['ms_per_step'], d.get('unpaired_ms_per_step', 0), h.get('ms_per_step', 0)))"; }
{
run "blocking update" ""
run "asynchronous update" "--async-update"
run "blocking update" ""
run "asynchronous update" "--async-update"
timeout 300 python tools/solve_c4_lp.py 2>/dev/null | grep -A1 HSD
} > gpurun_out/r03_async_update.txt 2>&1
cat gpurun_out/r03_async_update.txt
