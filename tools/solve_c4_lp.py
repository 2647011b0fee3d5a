"""End-to-end: a feasible, bounded LP on the constraint matrix of BASELINE configs[3] (64 blocks x (5000 x 10000) + 1000 linking
rows, m = 321 000, n = 640 000) solved by the device-resident HSD loop (tulip.jl_amd/hsd_device.py): analyse once, then one
update! + 3..6 solve! per iteration, iterate in HBM.  Prints status, iterations, objectives, residual measures and wall times."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import tulip_jl_amd as tk   # noqa: E402
from tulip_jl_amd.hsd_device import DeviceHSD   # noqa: E402
from tulip_jl_amd.mpc_device import DeviceMPC   # noqa: E402
from workloads import block_angular_lp   # noqa: E402

NSHARDS = int(os.environ.get("NSHARDS", "1"))       # > 1: one tlpk_create_multi handle, NSHARDS shards -- on this box's single GPU unless DEVICES lists ordinals
MULTI = dict(ngpus=NSHARDS, devices=[int(d) for d in os.environ.get("DEVICES", ",".join(["0"] * NSHARDS)).split(",")]) if NSHARDS > 1 else {}
HEADLINE = os.environ.get("HEADLINE") == "1"       # the north-star instance: 100 blocks x (20 000 inequality rows x 10 000 vars) + 1000 linking rows
A, row_block = block_angular_lp(100, 20000, 10000, 1000, 4, 0.5, ineq=True) if HEADLINE else block_angular_lp()
m, n = A.shape
rng = np.random.default_rng(20260927)
xs = rng.uniform(0.0, 1.0, n) * (rng.random(n) < 0.6)          # a vertex-ish feasible point (slack columns included)
b = A @ xs
ys = rng.standard_normal(m)
zs = rng.uniform(0.0, 1.0, n) * (xs == 0.0)                    # complementary slack
c = A.T @ ys + zs
l = np.zeros(n); u = np.full(n, np.inf)
ALGS = os.environ.get("ALGS", "HSD,MPC").split(",")
for name, cls in (("HSD", DeviceHSD), ("MPC", DeviceMPC)):
    if name not in ALGS:
        continue
    t0 = time.perf_counter()
    opt = cls(A, b, c, l, u, device=0, row_block=row_block, **MULTI)
    t_setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    opt.optimize()
    t_opt = time.perf_counter() - t0
    print(f"{name}: status {opt.status}  iterations {opt.niter}  primal {opt.primal_objective:.10e}  dual {opt.dual_objective:.10e}  "
          f"known optimum {float(c @ xs):.10e}  rho (p, d, gap) = {tuple(float('%.2e' % r) for r in opt.rho)}")
    print(f"     setup (analyse + upload) {t_setup:.2f} s   optimize {t_opt:.2f} s = {1e3 * t_opt / max(opt.niter, 1):.1f} ms per iteration "
          f"({opt.timers['n_update']} update!, {opt.timers['n_solve']} solve!, {opt.timers['n_bump']} regularisation bumps)")
    opt.kkt.close()
