#!/bin/bash
# Round 5, GPU session W: IPM-level parity at benchmark scale with the round-5 kernels (tools/ipm_parity_at_scale.py: the restated HSD / MPC loops, KKT backend swapped between
# the HIP library and the supernodal CPU comparator): north-star LP (100 blocks, m = 2 001 000) HSD; C4-matrix LP (64 blocks) HSD and MPC.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
HEADLINE=1 ALGS=HSD timeout 900 python tools/ipm_parity_at_scale.py > gpurun_out/r05_ipm_parity_headline_hsd.txt 2>&1
NB=64 ALGS=HSD,MPC timeout 900 python tools/ipm_parity_at_scale.py > gpurun_out/r05_ipm_parity_c4.txt 2>&1
cut -c1-260 gpurun_out/r05_ipm_parity_headline_hsd.txt gpurun_out/r05_ipm_parity_c4.txt
