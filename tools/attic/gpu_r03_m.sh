#!/bin/bash
# Round 3, GPU session M: device-resident HSD / MPC loops on multi-device handles.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hsd_device.py tests/test_mpc_device.py -m gpu -q --maxfail=8 -s > gpurun_out/r03_m_pytest.txt 2>&1
tail -40 gpurun_out/r03_m_pytest.txt
