"""Does MPC's late-phase trouble on the inequality block-angular LPs come from the HIP backend or from the algorithm?
Same LP, same host-vector MPC loop (tests/ipm_harness.py), HIP backend vs the CPU oracle backend."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tulip_jl_amd as tk   # noqa
from workloads import block_angular_lp
from ipm_harness import MPC, HSD, HipBackend, OracleBackend, IPMData
nb = int(os.environ.get("NB", "6"))
A, row_block = block_angular_lp(nb, 2000, 1000, 100, 4, 0.5, ineq=True)
m, n = A.shape
rng = np.random.default_rng(20260927)
xs = rng.uniform(0.0, 1.0, n) * (rng.random(n) < 0.6)
b = A @ xs; ys = rng.standard_normal(m); zs = rng.uniform(0.0, 1.0, n) * (xs == 0.0); c = A.T @ ys + zs
d = IPMData(); d.A = A.tocsc(); d.b = b; d.c = c; d.c0 = 0.0; d.objsense = True
d.l = np.zeros(n); d.u = np.full(n, np.inf); d.lflag = np.isfinite(d.l); d.uflag = np.isfinite(d.u)
d.lz = np.where(d.lflag, d.l, 0.0); d.uz = np.where(d.uflag, d.u, 0.0); d.nrow, d.ncol, d.nvar = m, n, n
print("LP", m, n, "known optimum", float(c @ xs))
for alg in ((MPC,) if os.environ.get("ONLY_MPC") else (MPC, HSD)):
    for name, be in (("HIP", lambda: HipBackend(d.A, device=0, row_block=row_block)), ("oracle", lambda: OracleBackend(d.A))):
        t0 = time.perf_counter()
        ipm = alg(d, be(), None).optimize()
        print(f"{alg.__name__:4s} {name:7s} status {ipm.status:20s} iterations {ipm.niter:3d} bumps {ipm.timers['n_bump']:3d} primal {ipm.primal_objective:.9e} rho {tuple(float('%.1e' % r) for r in ipm.rho)}  {time.perf_counter()-t0:.1f} s")
