#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06u
TLPK_TIMING=1 timeout 600 python tools/analyse_phases.py > ${O}_analyse_c4.txt 2>&1
HEADLINE=1 TLPK_TIMING=1 timeout 600 python tools/analyse_phases.py > ${O}_analyse_headline.txt 2>&1
tail -34 ${O}_analyse_c4.txt; tail -34 ${O}_analyse_headline.txt
python - <<'P'
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from workloads import block_angular_lp
import tulip_jl_amd as tk
A, rb = block_angular_lp()
for rep in range(3):
    t0 = time.perf_counter(); kkt = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb)); t1 = time.perf_counter()
    st = kkt.stats(); print("setup with device %.3f s, ms_analyse %.1f" % (t1 - t0, st.get("ms_analyse")), {k: v for k, v in st.items() if "ms_" in k}); kkt.close()
P
