"""Time tlpk_update on the C4 matrix with a library whose numbers may be WRONG (kernel ablations: the factorisation then reports a failed pivot, which this script
ignores -- the launches are the same).  TLPK_LIB selects the build."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tulip_jl_amd as tk
from workloads import block_angular_lp, kernel_inputs

A, rb = block_angular_lp()[:2]
kkt = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb))
th, rp, rd = kernel_inputs(kkt.m, kkt.n)[:3]
ts = []
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        tk.update(kkt, th, rp, rd)
    except Exception as e:
        pass
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("TLPK_LIB", "default")[-16:], "update ms (host-pointer ABI, incl. 3 vector uploads):", " ".join("%.2f" % t for t in ts[2:]), "median %.2f" % float(np.median(ts[2:])))
