#!/bin/bash
# Round 3, GPU session A: IPM-level parity at benchmark scale (HIP vs CPU supernodal backend) and the MPC PosDef diagnosis.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
nproc > gpurun_out/r03_host.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r03_host.txt
HEADLINE=1 EVENTS=3 timeout 900 python tools/mpc_posdef_diagnosis.py > gpurun_out/r03_mpc_posdef_diagnosis_headline.txt 2>&1
tail -5 gpurun_out/r03_mpc_posdef_diagnosis_headline.txt
NB=64 timeout 900 python tools/ipm_parity_at_scale.py > gpurun_out/r03_ipm_parity_c4.txt 2>&1
tail -7 gpurun_out/r03_ipm_parity_c4.txt
HEADLINE=1 NB=8 ALGS=MPC timeout 600 python tools/ipm_parity_at_scale.py > gpurun_out/r03_ipm_parity_headline_8blocks_mpc.txt 2>&1
tail -4 gpurun_out/r03_ipm_parity_headline_8blocks_mpc.txt
