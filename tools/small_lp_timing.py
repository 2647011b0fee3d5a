"""Per-step time of the KKT backend on the small / mid-size general sparse LPs (BASELINE configs C2 / C5 classes):
1 update! + 4 solve! through the device-pointer-free Python mirror (host vectors), and the number of launches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import tulip_jl_amd as tk
from tulip_jl_amd.problem import read_free_mps, standard_form
from lp_generators import multicommodity_lp
from helpers import ipm_like_data

def run(name, lp):
    d = standard_form(lp)
    A = d.A
    m, n = A.shape
    t0 = time.perf_counter(); kkt = tk.setup(A, tk.K1(), tk.Backend(device=0)); ta = time.perf_counter() - t0
    th, rp, rd, xp, xd = ipm_like_data(m, n, 1)
    dx = np.zeros(n); dy = np.zeros(m)
    for _ in range(3):
        tk.update(kkt, th, rp, rd); tk.solve(dx, dy, kkt, xp, xd)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        tk.update(kkt, th, rp, rd)
    tu = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        tk.solve(dx, dy, kkt, xp, xd)
    ts = (time.perf_counter() - t0) / reps
    st = kkt.stats()
    kkt.set_profile(True)
    tk.update(kkt, th, rp, rd); tk.solve(dx, dy, kkt, xp, xd)
    kt = kkt.kernel_times()
    kkt.set_profile(False)
    print("   per class (one update + one solve, every launch alone): " + ", ".join(f"{k} {v['ms']:.2f} ms/{v['launches']}" for k, v in kt.items() if v["launches"]))
    print("   flops_chol %.3e  max_front %d" % (st["flops_chol"], st["max_front"]))
    print(f"{name}: m={m} n={n} nnzL={st['nnzL']} supernodes={st['n_supernodes']} levels={st['n_levels']} launches update/solve={st['launches_update']}/{st['launches_solve']} "
          f"analyse {ta*1e3:.1f} ms  update {tu*1e3:.3f} ms  solve {ts*1e3:.3f} ms  step(1+4) {(tu+4*ts)*1e3:.3f} ms")

g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
run("stair25 (25fv47 class)", read_free_mps(os.path.join(g, "stair25.mps")))
run("pdseq20 (pds-20 class)", multicommodity_lp())
