"""Round 5: Mehrotra's predictor-corrector on the NORTH-STAR LP (100 blocks x (20 000 inequality rows x 10 000 vars) + 1000 linking
rows, m = 2 001 000, n = 3 000 000) with the CPU KKT backend ONLY -- the CHOLMOD-class supernodal comparator
(oracle/k1_supernodal.c behind tests/ipm_harness.py: SupernodalBackend), no GPU anywhere.

Why: on the HIP backend this LP stops at MPC's iteration limit with ~36 regularisation bumps (profiles/r04_mpc_levers.txt), and the
claim "the stall belongs to MPC's regularisation schedule (/root/reference/src/IPM/MPC/step.jl:28-51), not to the HIP factorisation"
rested on three re-factorised failing matrices and on the 8-block LP.  This is the direct evidence: the same restated loop, the same
LP, a CPU Cholesky.

    python tools/mpc_cpu_northstar.py [NB=100] > profiles/r05_mpc_cpu_northstar.txt

Prints one line per iteration (objectives, residual norms, mu, regularisations, bumps so far) and a summary line.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from ipm_harness import HSD, MPC, Options, SupernodalBackend   # noqa: E402
from ipm_parity_at_scale import make_lp   # noqa: E402


def main():
    nb = int(os.environ.get("NB", "100"))
    alg = os.environ.get("ALG", "MPC")
    threads = int(os.environ.get("THREADS", "0"))
    d, row_block, known = make_lp(nb, headline=os.environ.get("HEADLINE", "1") == "1")
    print(f"# LP: north-star shape, {nb} blocks, m = {d.nrow}, n = {d.ncol}, constructed optimum {known:.10e}", flush=True)
    t0 = time.perf_counter()
    be = SupernodalBackend(d.A, threads=threads, row_block=row_block)
    print(f"# CPU supernodal backend ready in {time.perf_counter() - t0:.1f} s ({be.o.threads} threads)", flush=True)
    cls = MPC if alg == "MPC" else HSD
    ipm = cls(d, be, Options())
    # one line per iteration, flushed: the run takes about an hour on 8 cores
    step = ipm.compute_step
    def logged_step():
        t = time.perf_counter()
        b0 = ipm.timers["n_bump"]
        step()
        it, po, do, rp, rd, _, mu = ipm.log[-1]
        print(f"{it:4d}  primal {po:+.10e}  dual {do:+.10e}  |rp| {rp:8.2e} |rd| {rd:8.2e}  mu {mu:8.2e}  "
              f"regP {float(ipm.regP.max()):.1e} regD {float(ipm.regD.max()):.1e}  bumps {ipm.timers['n_bump']:3d} (+{ipm.timers['n_bump'] - b0})  "
              f"alpha_p {getattr(ipm, 'alpha_p', float('nan')):.3f} alpha_d {getattr(ipm, 'alpha_d', float('nan')):.3f}  {time.perf_counter() - t:.1f} s", flush=True)
    ipm.compute_step = logged_step
    t0 = time.perf_counter()
    ipm.optimize()
    dt = time.perf_counter() - t0
    rel = lambda x, y: abs(x - y) / (1.0 + abs(y))      # noqa: E731
    print(f"{alg} supernodal(CPU) status {ipm.status} iterations {ipm.niter} bumps {ipm.timers['n_bump']} update! {ipm.timers['n_update']} "
          f"solve! {ipm.timers['n_solve']} primal {ipm.primal_objective:+.12e} dual {ipm.dual_objective:+.12e} "
          f"rho {tuple(float('%.2e' % r) for r in ipm.rho)} rel. gap to the constructed optimum {rel(ipm.primal_objective, known):.2e}  {dt:.0f} s "
          f"(factorisations {ipm.timers['Factorization']:.0f} s, solves {ipm.timers['KKT']:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
