#!/bin/bash
# Round 5, session K2: diagnostics of the two retry-loop tests with the row-oriented potrf_block_dpp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cat > /tmp/diag.py <<'P'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from test_hsd_device import device_hsd, GOLDEN, read_free_mps
lp = read_free_mps(os.path.join(GOLDEN, "bump.mps"))
dev, sd = device_hsd(lp)
print("mode", os.environ.get("TLPK_POTRF_MODE"), dict(dev.timers), "n_bump", dev.timers["n_bump"], "status", sd["status"], "niter", dev.niter, "z", sd.get("z_primal"), "n_update", dev.timers["n_update"])
P
for m in 4 3 2; do TLPK_POTRF_MODE=$m timeout 120 python /tmp/diag.py 2>&1 | tail -3; done | tee gpurun_out/r05k2_bump.txt
