"""hipGraph replay of the block-angular (four-stream) schedule WITHOUT torch in the process: host-pointer ABI only.  TLPK_GRAPH=2 forces the capture.
    TLPK_GRAPH=2 python tools/graph_without_torch.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tulip_jl_amd as tk
assert "torch" not in sys.modules, "torch got imported"
from workloads import block_angular_lp, kernel_inputs
A, rb = block_angular_lp(int(os.environ.get("NB", "16")))
m, n = A.shape
th, rp, rd, xp, xd = kernel_inputs(m, n, 7, "mid")
kkt = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb))
for it in range(4):
    t0 = time.perf_counter()
    tk.update(kkt, th, rp, rd)
    dx = np.empty(n); dy = np.empty(m)
    tk.solve(dx, dy, kkt, xp, xd)
    print("iteration", it, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), "residual %.2e" % np.abs(A @ dx + rd * dy - xp).max(), flush=True)
print("torch imported:", "torch" in sys.modules)
