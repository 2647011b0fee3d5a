"""Host analyse (KKT.setup without a device) of the bench shapes with TLPK_TIMING=1: where the milliseconds of `config.ms_analyse` go.  HEADLINE=1: the north-star shape."""
import sys, os, time, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from workloads import block_angular_lp
import tulip_jl_amd as tk
HEAD = os.environ.get("HEADLINE") == "1"
A, rb = (block_angular_lp(100, 20000, 10000, 1000, 4, 0.5, ineq=True) if HEAD else block_angular_lp())
for rep in range(2):
    t0 = time.perf_counter()
    kkt = tk.setup(A, tk.K1(), tk.Backend(device=-1, row_block=rb))
    print("setup (analyse only) %.3f s" % (time.perf_counter() - t0), kkt.stats().get("ms_analyse"))
