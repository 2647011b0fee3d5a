#!/bin/bash
# Round 5, GPU session Z: the device-resident MPC loop alone (57.2 ms per iteration on the C4-matrix LP against 43.0 + 3.4 x 2.5 = 51.5 of KKT work): kernel and copy statistics
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05z
ALGS=MPC timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --stats --output-format csv -d ${O}_prof -- python tools/solve_c4_lp.py > ${O}_lp.log 2>&1
grep "MPC\|setup" ${O}_lp.log | tail -2
D=$(ls -d ${O}_prof/*/ | head -1)
head -45 ${D}*kernel_stats.csv | cut -c1-150 > ${O}_kernel_stats_mpc.csv
head -25 ${D}*hip_api_stats.csv | cut -c1-150 > ${O}_hip_stats_mpc.csv
cat ${D}*memory_copy_stats.csv > ${O}_memcpy_stats_mpc.csv
rm -rf ${O}_prof
grep -i "ipm\|mpc\|rocclr" ${O}_kernel_stats_mpc.csv | cut -c1-140; cat ${O}_hip_stats_mpc.csv; cat ${O}_memcpy_stats_mpc.csv
