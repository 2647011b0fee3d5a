#!/bin/bash
# Round 5, GPU session Y: the reductions of the device-resident interior-point loops -- k_ipm_finalize with one wave per slot (was one thread per slot over 1024 blocks: 363 us
# per call), k_ipm_res_rows with 8 lanes per row (was 441 us): end-to-end LPs and the device-loop tests.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05y
timeout 300 python tools/solve_c4_lp.py 2>&1 | tail -4 | tee ${O}_c4_lp.txt
HEADLINE=1 timeout 400 python tools/solve_c4_lp.py 2>&1 | tail -4 | tee ${O}_headline_lp.txt
timeout 900 python -m pytest tests/test_hsd_device.py tests/test_mpc_device.py tests/test_presolve.py tests/test_lp_configs.py -m gpu -q 2>&1 | tail -4 | tee ${O}_pytest.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_prof -- python tools/solve_c4_lp.py > ${O}_lp.log 2>&1
grep "k_ipm_finalize\|k_ipm_res_rows" $(ls ${O}_prof/*/*kernel_stats.csv | head -1) | cut -c1-140 | tee ${O}_ipm_kernels.txt; rm -rf ${O}_prof
