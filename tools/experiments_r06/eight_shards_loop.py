"""Eight shards of one job on ONE GPU, N updates on one handle: how many come back with TLPK_INTERNAL (a dependency-driven launch gave up waiting).  TLPK_LIB selects the build,
TLPK_CHAIN_RETRY=0 counts the raw events."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tulip_jl_amd as tk
from workloads import block_angular_lp, kernel_inputs
N = int(os.environ.get("N", "2000"))
A, rb = block_angular_lp(32)
m, n = A.shape
th, rp, rd, xp, xd = kernel_inputs(m, n, 11, "mid")
kkt = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb, ngpus=8, devices=[0] * 8))
bad = 0; t0 = time.time()
for it in range(N):
    try:
        tk.update(kkt, th, rp, rd)
    except RuntimeError as e:
        bad += 1
        print("update", it, "->", str(e)[:90], flush=True)
print(os.environ.get("TLPK_LIB", "default")[-24:], f"{bad} of {N} updates gave up; {(time.time() - t0) / N * 1e3:.1f} ms per update", flush=True)
