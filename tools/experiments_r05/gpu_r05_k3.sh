#!/bin/bash
# Round 5, session K3: the bump LP -- at every failed factorisation of the device-resident HSD loop, repeat the same factorisation with the other block kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cat > /tmp/diag.py <<'P'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
os.environ["TLPK_POTRF_DYN"] = "1"
BASE = sys.argv[1]; OTHER = sys.argv[2]
os.environ["TLPK_POTRF_MODE"] = BASE
import tulip_jl_amd as tk
from tulip_jl_amd.hsd_device import DeviceHSD
from tulip_jl_amd.kkt import PosDefException
from tulip_jl_amd.problem import read_free_mps, standard_form
lp = read_free_mps("tests/golden/bump.mps"); d = standard_form(lp)
opt = DeviceHSD(d.A, d.b, d.c, d.l, d.u, c0=d.c0, objsense_min=d.objsense, device=0, overlap_root=False)
L = opt.L
orig = L.tlpk_ipm_factor
class W:
    def __call__(self, h, regP, regD):
        rc = orig(h, regP, regD)
        os.environ["TLPK_POTRF_MODE"] = OTHER
        rc2 = orig(h, regP, regD)
        os.environ["TLPK_POTRF_MODE"] = BASE
        rc3 = orig(h, regP, regD)
        st = opt.kkt.stats()
        print("iter %2d regP %.1e: mode %s rc %d | mode %s rc %d | mode %s again rc %d" % (opt.niter, regP, BASE, rc, OTHER, rc2, BASE, rc3), flush=True)
        return rc3
class LW:
    def __getattr__(self, k):
        return W() if k == "tlpk_ipm_factor" else getattr(L, k)
opt.L = LW()
opt.optimize()
print("base mode", BASE, "status", opt.status, "niter", opt.niter, dict(opt.timers))
P
timeout 200 python /tmp/diag.py 3 2 2>&1 | tail -40 | tee gpurun_out/r05k3_bump.txt
timeout 200 python /tmp/diag.py 2 3 2>&1 | tail -40 | tee -a gpurun_out/r05k3_bump.txt
