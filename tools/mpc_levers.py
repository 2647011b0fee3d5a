"""MPC on the north-star LP family stops at its iteration limit with dozens of regularisation bumps on the K1 backend (round 3: the failing
matrices A D A' + sqrt(eps) I are indefinite to rounding on the CPU backend too).  The two levers the product owns, tried here on the same
LP (tools/solve_c4_lp.py: the north-star shape, NB blocks): iterative refinement of the K1 solves (refine = 1, 2; the reference leaves it as a
TODO, spd.jl:68) and the augmented system K2 (the reference's DEFAULT linear system for Float64, KKT.jl:134-141; better conditioned: no
A D A' product).  Prints iterations / bumps / status / objectives per variant.
    NB=8 python tools/mpc_levers.py        NB=100 python tools/mpc_levers.py"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import tulip_jl_amd as tk   # noqa: E402
from tulip_jl_amd.hsd_device import DeviceHSD   # noqa: E402
from tulip_jl_amd.mpc_device import DeviceMPC   # noqa: E402
from workloads import block_angular_lp   # noqa: E402

NB = int(os.environ.get("NB", "8"))
SHAPE = os.environ.get("SHAPE", "headline")
A, row_block = block_angular_lp(NB, 20000, 10000, 1000, 4, 0.5, ineq=True) if SHAPE == "headline" else block_angular_lp(NB)
m, n = A.shape
rng = np.random.default_rng(20260927)
xs = rng.uniform(0.0, 1.0, n) * (rng.random(n) < 0.6)
b = A @ xs
ys = rng.standard_normal(m)
zs = rng.uniform(0.0, 1.0, n) * (xs == 0.0)
c = A.T @ ys + zs
l = np.zeros(n); u = np.full(n, np.inf)
print(f"LP: {SHAPE} shape, {NB} blocks, m = {m}, n = {n}, known optimum {float(c @ xs):.10e}")
variants = [("MPC K1", DeviceMPC, dict(system="K1")), ("MPC K1 refine=1", DeviceMPC, dict(system="K1", refine=1)),
            ("MPC K1 refine=2", DeviceMPC, dict(system="K1", refine=2)), ("MPC K2", DeviceMPC, dict(system="K2")),
            ("HSD K1", DeviceHSD, dict(system="K1")), ("HSD K2", DeviceHSD, dict(system="K2"))]
only = os.environ.get("ONLY")
for name, cls, kw in variants:
    if only and only not in name:
        continue
    try:
        t0 = time.perf_counter()
        opt = cls(A, b, c, l, u, device=0, row_block=row_block, **kw)
        t_setup = time.perf_counter() - t0
        t0 = time.perf_counter()
        opt.optimize()
        t_opt = time.perf_counter() - t0
        print(f"{name:16s}: {opt.status:18s} iterations {opt.niter:3d}  bumps {opt.timers['n_bump']:3d}  update! {opt.timers['n_update']:3d}  solve! {opt.timers['n_solve']:3d}  "
              f"primal {opt.primal_objective:.10e}  dual {opt.dual_objective:.10e}  rho {tuple(float('%.1e' % r) for r in opt.rho)}  "
              f"setup {t_setup:.1f} s  optimize {t_opt:.1f} s", flush=True)
        opt.kkt.close()
    except Exception as e:
        print(f"{name:16s}: FAILED {type(e).__name__}: {e}", flush=True)
