#!/usr/bin/env python3
"""CPU comparator (oracle/k1_supernodal.c) on the host of the GPU box: where does a solve spend its time, and how do the threading modes of
the triangular solves compare (K1SN_SOLVE_TEAM=0|1|2, K1SN_TRACE=1 prints one line per tree level).  No GPU work.
    python tools/cpu_solve_probe.py c4|headline [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tulip_jl_amd as tk
from oracle_binding import SupernodalK1
from workloads import block_angular_lp, kernel_inputs

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, e)
print("sched_getaffinity:", len(os.sched_getaffinity(0)), "cpus")
A, rb = block_angular_lp() if which == "c4" else block_angular_lp(100, 20000, 10000, 1000, 4, 0.5, ineq=True)
m, n = A.shape
full = tk.setup(A, tk.K1(), tk.Backend(device=-1, row_block=rb))
st = full.stats()
th, rp, rd, xp, xd = kernel_inputs(m, n, 7, "mid")
sn = SupernodalK1(A, full, threads=threads)
sn.update(th, rp, rd)
t0 = time.perf_counter(); sn.update(th, rp, rd); tu = time.perf_counter() - t0
print(f"{which}: update {tu:.2f} s on {sn.threads} threads = {st['flops_chol'] / tu / 1e9:.0f} GFLOP/s", flush=True)
for mode in ("1", "0", "2"):
    os.environ["K1SN_SOLVE_TEAM"] = mode
    os.environ.pop("K1SN_TRACE", None)
    sn.solve(xp, xd)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); dx, dy = sn.solve(xp, xd); ts.append(time.perf_counter() - t0)
    print(f"K1SN_SOLVE_TEAM={mode}: solve {min(ts):.3f} s = {2 * 8 * st['nnzL_stored'] / min(ts) / 1e9:.0f} GB/s of factor traffic; residual {float(np.abs(A @ dx + rd * dy - xp).max()):.2e}", flush=True)
    os.environ["K1SN_TRACE"] = "1"
    sys.stderr.flush(); sn.solve(xp, xd); sys.stderr.flush()
