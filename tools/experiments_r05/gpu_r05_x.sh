#!/bin/bash
# Round 5, GPU session X: where do the 58 ms of a device-resident HSD iteration on the C4-matrix LP go (bench step: 50.8 ms for 1 update + 4 right-hand sides)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05x
ALGS=HSD timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d ${O}_prof -- python tools/solve_c4_lp.py > ${O}_lp.log 2>&1
cp $(ls ${O}_prof/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_hsd.csv
ls ${O}_prof/*/ | head; cp $(ls ${O}_prof/*/*memory_copy_stats.csv 2>/dev/null | head -1) ${O}_memcpy_stats_hsd.csv 2>/dev/null
rm -rf ${O}_prof
tail -4 ${O}_lp.log; head -40 ${O}_kernel_stats_hsd.csv | cut -c1-160; cat ${O}_memcpy_stats_hsd.csv 2>/dev/null | head
