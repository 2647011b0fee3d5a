"""Why does Mehrotra's predictor-corrector collect `PosDefException` bumps on the north-star instance?

Runs the device-resident MPC loop (tulip.jl_amd/mpc_device.py; caller restated from /root/reference/src/IPM/MPC/step.jl:28-51)
on the LP of tools/solve_c4_lp.py and, at each of the first EVENTS failed factorisations (TLPK_NOT_POSDEF from
tlpk_ipm_factor), pulls the iterate off the device, rebuilds (theta_inv, regP, regD) on the host and factorises THE SAME DATA
 * again on the HIP library through the host-pointer ABI (is the failure reproducible, which column), and
 * on the CHOLMOD-class CPU comparator (oracle/k1_supernodal.c: dpotrf / dtrsm / dsyrk, same ordering and supernodes),
and reports for both: success or the failing column, the smallest pivot L_jj^2 seen, the pivot the CPU computes at the column
the HIP factorisation rejected.  If the CPU succeeds where HIP fails, the pivot path of the HIP kernels is to blame; if both
fail, the matrix is numerically indefinite at this regularisation and the bump belongs to the algorithm's schedule.

    HEADLINE=1 python tools/mpc_posdef_diagnosis.py      # north-star instance (100 blocks), on the GPU box
    NB=8 HEADLINE=1 ... / NB=16 ...                      # smaller instances of the same families
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tulip_jl_amd as tk   # noqa: E402
from tulip_jl_amd.mpc_device import DeviceMPC   # noqa: E402
from tulip_jl_amd.kkt import PosDefException   # noqa: E402
from tulip_jl_amd import _lib   # noqa: E402
from ipm_parity_at_scale import make_lp   # noqa: E402
from oracle_binding import OraclePosDefError, SupernodalK1   # noqa: E402

HEADLINE = os.environ.get("HEADLINE") == "1"
NB = int(os.environ.get("NB", "100" if HEADLINE else "64"))
EVENTS = int(os.environ.get("EVENTS", "3"))
MAXIT = int(os.environ.get("MAXIT", "100"))

d, row_block, known = make_lp(NB, HEADLINE)
m, n = d.nrow, d.ncol
print(f"LP: {'headline' if HEADLINE else 'C4'} shape, {NB} blocks, m = {m}, n = {n}, known optimum {known:.10e}", flush=True)

t0 = time.perf_counter()
opt = DeviceMPC(d.A, d.b, d.c, d.l, d.u, device=0, row_block=row_block)
print(f"HIP setup {time.perf_counter() - t0:.2f} s", flush=True)
sym = tk.setup(d.A, tk.K1(), tk.Backend(device=-1, row_block=row_block))          # analyse-only twin: symbolic arrays for the CPU comparator
cpu = SupernodalK1(d.A, sym)
host = tk.setup(d.A, tk.K1(), tk.Backend(device=0, row_block=row_block))          # a second HIP handle, driven through the host-pointer ABI
print(f"CPU comparator: {cpu.threads} threads", flush=True)

events = []
lf, uf = np.isfinite(d.l), np.isfinite(d.u)


def theta_from_device():
    xl, xu, zl, zu = (opt._get(w, n) for w in (1, 2, 3, 4))
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(lf, zl / xl, 0.0) + np.where(uf, zu / xu, 0.0)


def factor_both(th, regP, regD, tag):
    rp, rd = np.full(n, regP), np.full(m, regD)
    rec = {"tag": tag, "regP": regP, "regD": regD, "theta_min": float(th.min()), "theta_max": float(th.max())}
    try:
        tk.update(host, th, rp, rd); rec["hip"] = "ok"; rec["hip_col"] = -1
    except PosDefException:
        rec["hip"] = "NOT_POSDEF"; rec["hip_col"] = int(host.stats()["fail_col"])
    t1 = time.perf_counter()
    try:
        cpu.update(th, rp, rd); rec["cpu"] = "ok"; rec["cpu_col"] = -1
    except OraclePosDefError as e:
        rec["cpu"] = "NOT_POSDEF"; rec["cpu_col"] = int(e.args[0])
    rec["cpu_s"] = time.perf_counter() - t1
    dg = cpu.diag()
    if rec["cpu"] == "ok":
        piv = dg * dg
        rec["cpu_min_pivot"] = float(piv.min()); rec["cpu_argmin"] = int(piv.argmin()); rec["cpu_max_pivot"] = float(piv.max())
        if rec["hip_col"] >= 0:
            rec["cpu_pivot_at_hip_col"] = float(piv[rec["hip_col"]])
    print(f"  [{tag}] regP {regP:.3e} regD {regD:.3e} theta_inv in [{rec['theta_min']:.2e}, {rec['theta_max']:.2e}]  HIP: {rec['hip']}"
          + (f" at column {rec['hip_col']}" if rec["hip_col"] >= 0 else "")
          + f"   CPU: {rec['cpu']}" + (f" at column {rec['cpu_col']}" if rec["cpu_col"] >= 0 else "")
          + (f"  min pivot {rec['cpu_min_pivot']:.3e} (column {rec['cpu_argmin']}), max {rec['cpu_max_pivot']:.3e}" if "cpu_min_pivot" in rec else "")
          + (f", CPU pivot at HIP's column {rec['cpu_pivot_at_hip_col']:.3e}" if "cpu_pivot_at_hip_col" in rec else "")
          + f"   ({rec['cpu_s']:.1f} s CPU)", flush=True)
    return rec


# the loop of DeviceMPC.compute_step with a hook at the failed factorisation
L = opt.L
orig_factor = L.tlpk_ipm_factor


def hooked_step():
    """DeviceMPC.compute_step, with the data of the first failed factorisations examined on both backends."""
    it = opt.niter
    pre_regP, pre_regD = min(max(opt.regP / 10, 1.4901161193847656e-08), 1.0), min(max(opt.regD / 10, 1.4901161193847656e-08), 1.0)
    rc = orig_factor(opt.kkt._h, pre_regP, pre_regD)
    if rc == _lib.NOT_POSDEF and len(events) < EVENTS:
        col = int(opt.kkt.stats()["fail_col"])
        print(f"iteration {it}: tlpk_ipm_factor -> NOT_POSDEF at column {col} (mu = {opt.mu:.3e}, rho = {tuple(float('%.1e' % r) for r in opt.rho)})", flush=True)
        th = theta_from_device()
        ev = {"iteration": it, "device_col": col, "runs": [factor_both(th, pre_regP, pre_regD, "as failed")]}
        ev["runs"].append(factor_both(th, pre_regP * 100, pre_regD * 100, "bumped x100"))
        events.append(ev)
    DeviceMPC.compute_step(opt)        # the real step (repeats the factorisation with its retry loop)


opt.opt.IterationsLimit = MAXIT
t0 = time.perf_counter()
opt.niter = 0
opt.regP = opt.regD = 1.0
opt.compute_starting_point()
while True:
    opt.compute_residuals(); opt.update_solver_status()
    if opt.status in ("Trm_Optimal", "Trm_PrimalInfeasible", "Trm_DualInfeasible"):
        break
    if opt.niter >= MAXIT:
        opt.status = "Trm_IterationLimit"; break
    try:
        hooked_step()
    except PosDefException:
        opt.status = "Trm_NumericalProblem"; break
    opt.niter += 1
print(f"MPC on HIP: status {opt.status}  iterations {opt.niter}  bumps {opt.timers['n_bump']}  primal {opt.primal_objective:.10e}  "
      f"dual {opt.dual_objective:.10e}  rho {tuple(float('%.2e' % r) for r in opt.rho)}  {time.perf_counter() - t0:.1f} s", flush=True)
n_hip_only = sum(1 for e in events if e["runs"][0]["hip"] != "ok" and e["runs"][0]["cpu"] == "ok")
n_both = sum(1 for e in events if e["runs"][0]["hip"] != "ok" and e["runs"][0]["cpu"] != "ok")
print(f"verdict over {len(events)} examined failure(s): HIP fails / CPU succeeds on the same data: {n_hip_only};  both fail: {n_both}")
