"""End-to-end IPM checks through the restated HSD and MPC drivers (tests/ipm_harness.py).

CPU (always): the four example LPs of the reference with the answers its own test-suite asserts
(/root/reference/examples/*.jl, run by test/examples.jl) -- on the oracle backend: this is
BASELINE config C1 ("examples/optimal.jl tiny LP ... plumbing, no GPU"), and it pins the harness.
GPU (-m gpu): the same LPs and random feasible LPs on the HIP backend; iteration count, status,
objectives and residuals must match the CPU run (SURVEY.md section 8d parity protocol (ii))."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from ipm_harness import (HSD, LP, MPC, HipBackend, Options, OracleBackend, read_free_mps, solve_lp,
                         standard_form, _max_step_vec)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 100 * float(np.sqrt(np.finfo(float).eps))      # atol = rtol = 100*sqrt(eps) in examples/*.jl


def check_optimal(sol):          # examples/optimal.jl:37-62
    assert sol["status"] == "Trm_Optimal"
    assert sol["primal_status"] == sol["dual_status"] == "Sln_Optimal"
    assert abs(sol["z_primal"] - 1.5) <= TOL
    np.testing.assert_allclose(sol["x"], [0.5, 0.5], atol=TOL, rtol=TOL)
    np.testing.assert_allclose(sol["y"], [1.5, -0.5], atol=TOL, rtol=TOL)
    np.testing.assert_allclose(sol["s"], [0.0, 0.0], atol=TOL)


def check_infeasible(sol):       # examples/infeasible.jl:36-53
    assert sol["status"] == "Trm_PrimalInfeasible"
    assert sol["dual_status"] == "Sln_InfeasibilityCertificate" and sol["primal_status"] == "Sln_Unknown"
    y, s = sol["y"], sol["s"]
    assert y[0] + y[2] >= TOL
    assert abs(y[0] + y[1] + s[0]) <= TOL and abs(y[0] - y[1] + y[2] + s[1]) <= TOL
    assert s[0] >= -TOL and s[1] >= -TOL


def check_unbounded(sol, lp):    # examples/unbounded.jl:34-53
    assert sol["status"] == "Trm_DualInfeasible"
    assert sol["primal_status"] == "Sln_InfeasibilityCertificate" and sol["dual_status"] == "Sln_Unknown"
    x = sol["x"]
    assert x[0] >= -TOL and x[1] >= -TOL
    assert abs((lp.A @ x)[0]) <= TOL
    assert -x[0] - x[1] <= -TOL


def check_freevars(sol):         # examples/freevars.jl:35-57
    assert sol["status"] == "Trm_Optimal"
    assert abs(sol["z_primal"]) <= TOL
    x = sol["x"]
    assert 2 * x[0] + x[1] >= 2 - TOL and x[0] + 2 * x[1] >= 2 - TOL and x[0] + x[1] + x[2] >= -TOL
    np.testing.assert_allclose(sol["s"], 0.0, atol=TOL)


def run_examples(factory):
    out = {}
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_opt.mps"))
    out["opt"] = solve_lp(lp, factory); check_optimal(out["opt"][1])
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_inf.mps"))
    out["inf"] = solve_lp(lp, factory); check_infeasible(out["inf"][1])
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_ubd.mps"))
    out["ubd"] = solve_lp(lp, factory); check_unbounded(out["ubd"][1], lp)
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_freevars.mps"))
    out["free"] = solve_lp(lp, factory); check_freevars(out["free"][1])
    return out


def test_mps_reader_and_standard_form():
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_opt.mps"))
    np.testing.assert_array_equal(lp.A.toarray(), [[1, 1], [1, -1]])
    np.testing.assert_array_equal(lp.obj, [1, 2])
    np.testing.assert_array_equal(lp.uvar, [1, 1])
    assert (lp.lcon == lp.ucon).all() and list(lp.lcon) == [1, 0]
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_freevars.mps"))
    d = standard_form(lp)                     # 3 ">=" rows -> slack -1 each (ipmdata.jl:96-102)
    assert d.A.shape == (3, 6)
    np.testing.assert_array_equal(d.A[:, 3:].toarray(), -np.eye(3))
    assert not d.lflag[:3].any() and d.lflag[3:].all() and not d.uflag.any()


def test_reference_ipm_unit_facts():
    """test/IPM/HSD.jl:34-41 -- max_step_length known answers."""
    assert _max_step_vec(np.ones(2), np.ones(2)) == float("inf")
    assert _max_step_vec(np.ones(2), np.array([1.0, -1.0])) == 1.0
    assert _max_step_vec(np.ones(2), np.array([-2.0, -1.0])) == 0.5


def test_examples_on_cpu_backend_config_c1():
    out = run_examples(lambda A: OracleBackend(A))
    hsd = out["opt"][0]
    assert 3 <= hsd.niter <= 30
    assert hsd.timers["n_update"] == hsd.niter and hsd.timers["n_solve"] >= 3 * hsd.niter   # 1 factorise + 3..6 solves


def _mpc_unit_problem():
    """test/IPM/MPC.jl:45-63: min x1 - x2  s.t. x1 + x2 = 1, x1 - x2 = 0, 0 <= x <= 2."""
    A = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, -1.0]]))
    lp = LP(A, np.array([1.0, -1.0]), 0.0, np.array([1.0, 0.0]), np.array([1.0, 0.0]), np.zeros(2), np.full(2, 2.0))
    dat = standard_form(lp)
    return lp, MPC(dat, OracleBackend(dat.A))


def test_reference_mpc_unit_facts():
    """test/IPM/MPC.jl:65-90 (convergence at the known optimum) and :104-147 (residual identities)."""
    lp, ipm = _mpc_unit_problem()
    pt, d = ipm.pt, ipm.dat
    pt.x[:] = [0.5, 0.5]; pt.xl[:] = [0.5, 0.5]; pt.xu[:] = [1.5, 1.5]
    pt.y[:] = [0.0, 1.0]; pt.zl[:] = 0.0; pt.zu[:] = 0.0
    pt.tau, pt.kappa, pt.mu = 1.0, 0.0, 0.0
    ipm.compute_residuals()
    ipm.update_solver_status()
    assert ipm.status == "Trm_Optimal"
    x = pt.x[:] = [3.0, 5.0]; xl = pt.xl[:] = [1.0, 8.0]; xu = pt.xu[:] = [2.0, 1.0]
    y = pt.y[:] = [10.0, -2.0]; zl = pt.zl[:] = [2.0, 1.0]; zu = pt.zu[:] = [5.0, 7.0]
    x, xl, xu, y, zl, zu = (np.array(v) for v in (x, xl, xu, y, zl, zu))
    ipm.compute_residuals()
    A = d.A.toarray()
    np.testing.assert_allclose(ipm.rp, d.b - A @ x)
    np.testing.assert_allclose(ipm.rl, d.l - (x - xl))
    np.testing.assert_allclose(ipm.ru, d.u - (x + xu))
    np.testing.assert_allclose(ipm.rd, d.c - A.T @ y - zl + zu)
    assert ipm.rp_nrm == np.abs(ipm.rp).max() and ipm.rd_nrm == np.abs(ipm.rd).max()


def test_mpc_examples_on_cpu_backend():
    """test/examples.jl:8,17 run ex_optimal and ex_freevars with IPM_Factory = MPC; the other two
    example LPs exercise the certificate tests of MPC.jl:178-203.  The call pattern seen by the KKT
    backend: 1 update! + 2 solve! for the starting point, then 1 update! + >= 2 solve! per iteration."""
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_opt.mps"))
    ipm, sol = solve_lp(lp, lambda A: OracleBackend(A), algorithm="mpc"); check_optimal(sol)
    assert ipm.timers["n_update"] == ipm.niter + 1 and ipm.timers["n_solve"] >= 2 + 2 * ipm.niter
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_freevars.mps"))
    check_freevars(solve_lp(lp, lambda A: OracleBackend(A), algorithm="mpc")[1])
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_inf.mps"))
    assert solve_lp(lp, lambda A: OracleBackend(A), algorithm="mpc")[1]["status"] == "Trm_PrimalInfeasible"
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_ubd.mps"))
    assert solve_lp(lp, lambda A: OracleBackend(A), algorithm="mpc")[1]["status"] == "Trm_DualInfeasible"


def test_mpc_random_lp_matches_highs():
    from scipy.optimize import linprog
    lp = random_feasible_lp(30, 60, 2)
    ipm, sol = solve_lp(lp, lambda A: OracleBackend(A), algorithm="mpc")
    assert sol["status"] == "Trm_Optimal"
    ref = linprog(lp.obj, A_eq=lp.A, b_eq=lp.lcon, bounds=[(0, None)] * 60, method="highs")
    assert abs(sol["z_primal"] - ref.fun) <= 1e-6 * (1 + abs(ref.fun))


def random_feasible_lp(m, n, seed, ineq=False):
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=min(1.0, 4.0 / n), random_state=seed, format="csc",
                  data_rvs=rng.standard_normal)
    A = (A + sp.csc_matrix((np.ones(m), (np.arange(m), rng.integers(0, n, m))), shape=(m, n))).tocsc()
    x0 = rng.uniform(0.5, 1.5, n)
    y0 = rng.standard_normal(m); z0 = rng.uniform(0.1, 1.0, n)
    if ineq:
        y0 = -np.abs(y0)            # rows "a'x <= b" need multipliers <= 0 for a bounded LP
    c = A.T @ y0 + z0
    if ineq:
        s0 = rng.uniform(0.1, 1.0, m)
        return LP(A, c, 0.0, np.full(m, -np.inf), A @ x0 + s0, np.zeros(n), np.full(n, np.inf))
    b = A @ x0
    return LP(A, c, 0.0, b, b, np.zeros(n), np.full(n, np.inf))


def test_random_lp_matches_highs():
    from scipy.optimize import linprog
    lp = random_feasible_lp(30, 60, 1)
    hsd, sol = solve_lp(lp, lambda A: OracleBackend(A))
    assert sol["status"] == "Trm_Optimal"
    ref = linprog(lp.obj, A_eq=lp.A, b_eq=lp.lcon, bounds=[(0, None)] * 60, method="highs")
    assert abs(sol["z_primal"] - ref.fun) <= 1e-6 * (1 + abs(ref.fun))


@pytest.mark.gpu
def test_examples_on_hip_backend_match_cpu():
    cpu = run_examples(lambda A: OracleBackend(A))
    gpu = run_examples(lambda A: HipBackend(A, device=0))
    for k in cpu:
        hc, sc = cpu[k]; hg, sg = gpu[k]
        assert sg["status"] == sc["status"]
        assert abs(hg.niter - hc.niter) <= 1
        if sc["status"] == "Trm_Optimal":
            assert abs(sg["z_primal"] - sc["z_primal"]) <= 1e-8 * (1 + abs(sc["z_primal"]))
            assert abs(sg["z_dual"] - sc["z_dual"]) <= 1e-8 * (1 + abs(sc["z_dual"]))
            assert max(sg["rho"]) <= float(np.sqrt(np.finfo(float).eps))


@pytest.mark.gpu
@pytest.mark.parametrize("ineq", [False, True])
def test_random_lp_hip_vs_cpu_iterations_and_residuals(ineq):
    """same termination status, |delta niter| <= 1, objectives to 1e-8 relative,
    final rho_p, rho_d, rho_g <= sqrt(eps) -- HIP backend vs CPU oracle backend, same ordering."""
    lp = random_feasible_lp(300, 700, 5, ineq=ineq)
    hg, sg = solve_lp(lp, lambda A: HipBackend(A, device=0))
    perm = hg.kkt.kkt.perm()
    hc, sc = solve_lp(lp, lambda A: OracleBackend(A, perm))
    assert sg["status"] == sc["status"] == "Trm_Optimal"
    assert abs(hg.niter - hc.niter) <= 1
    assert abs(sg["z_primal"] - sc["z_primal"]) <= 1e-8 * (1 + abs(sc["z_primal"]))
    assert abs(sg["z_dual"] - sc["z_dual"]) <= 1e-8 * (1 + abs(sc["z_dual"]))
    assert max(sg["rho"]) <= float(np.sqrt(np.finfo(float).eps))
    from scipy.optimize import linprog
    if ineq:
        ref = linprog(lp.obj, A_ub=lp.A, b_ub=lp.ucon, bounds=[(0, None)] * 700, method="highs")
    else:
        ref = linprog(lp.obj, A_eq=lp.A, b_eq=lp.lcon, bounds=[(0, None)] * 700, method="highs")
    assert abs(sg["z_primal"] - ref.fun) <= 1e-6 * (1 + abs(ref.fun))


@pytest.mark.gpu
def test_block_angular_lp_end_to_end_on_hip():
    """A small block-angular LP through the whole IPM on the HIP backend with the row_block hook."""
    from helpers import block_angular
    A, row_block = block_angular(nblocks=4, mk=60, nk=150, m0=10, nnz_in=3, link_prob=0.5, seed=9)
    rng = np.random.default_rng(3)
    m, n = A.shape
    x0 = rng.uniform(0.5, 1.5, n); b = A @ x0
    c = A.T @ rng.standard_normal(m) + rng.uniform(0.1, 1.0, n)
    lp = LP(A, c, 0.0, b, b, np.zeros(n), np.full(n, np.inf))
    hg, sg = solve_lp(lp, lambda M: HipBackend(M, device=0, row_block=row_block))
    hc, sc = solve_lp(lp, lambda M: OracleBackend(M))
    assert sg["status"] == sc["status"] == "Trm_Optimal"
    assert abs(hg.niter - hc.niter) <= 1
    assert abs(sg["z_primal"] - sc["z_primal"]) <= 1e-8 * (1 + abs(sc["z_primal"]))


@pytest.mark.gpu
@pytest.mark.parametrize("ineq", [False, True])
def test_mpc_random_lp_hip_vs_cpu(ineq):
    """The MPC caller (starting-point update with theta_inv = 0 and regD = 1e-6, half-zero right-hand
    sides, separate primal/dual steps) on the HIP backend vs the CPU oracle backend."""
    lp = random_feasible_lp(300, 700, 11, ineq=ineq)
    hg, sg = solve_lp(lp, lambda A: HipBackend(A, device=0), algorithm="mpc")
    perm = hg.kkt.kkt.perm()
    hc, sc = solve_lp(lp, lambda A: OracleBackend(A, perm), algorithm="mpc")
    assert sg["status"] == sc["status"] == "Trm_Optimal"
    assert abs(hg.niter - hc.niter) <= 1
    assert abs(sg["z_primal"] - sc["z_primal"]) <= 1e-8 * (1 + abs(sc["z_primal"]))
    assert abs(sg["z_dual"] - sc["z_dual"]) <= 1e-8 * (1 + abs(sc["z_dual"]))
    assert max(sg["rho"]) <= float(np.sqrt(np.finfo(float).eps))


@pytest.mark.gpu
def test_mpc_examples_on_hip_backend():
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_opt.mps"))
    check_optimal(solve_lp(lp, lambda A: HipBackend(A, device=0), algorithm="mpc")[1])
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_freevars.mps"))
    check_freevars(solve_lp(lp, lambda A: HipBackend(A, device=0), algorithm="mpc")[1])
