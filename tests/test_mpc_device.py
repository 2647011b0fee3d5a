"""Device-resident Mehrotra predictor-corrector (`tlpk_mpc_*`): the loop of tests/ipm_harness.MPC
with every vector in HBM.  Validated against the host-vector path on the SAME backend (status, iteration count,
objectives, iterates side by side) and on the reference's example LPs."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import tulip_jl_amd as tk
from ipm_harness import MPC, HipBackend, read_free_mps, solve_lp, standard_form
from tulip_jl_amd import _lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SQRT_EPS = float(np.sqrt(np.finfo(float).eps))


def test_mpc_entry_points_fail_loudly_without_device():
    A = sp.csc_matrix(np.array([[1.0, 0, 1, 0], [0, 1, 0, 1]]))
    kkt = tk.setup(A, tk.K1(), tk.Backend(device=-1))
    L = _lib.lib()
    out = np.zeros(16)
    assert L.tlpk_mpc_start(kkt._h, _lib.as_pd(out)) == _lib.NO_DEVICE
    assert L.tlpk_mpc_newton(kkt._h, 0, 0.0, _lib.as_pd(out)) == _lib.NO_DEVICE
    assert L.tlpk_mpc_gap(None, 1.0, 1.0, _lib.as_pd(out)) == _lib.BADARG
    assert L.tlpk_mpc_advance(kkt._h, 1.0, 1.0, _lib.as_pd(out)) == _lib.NO_DEVICE


def device_mpc(lp):
    from tulip_jl_amd.mpc_device import DeviceMPC
    d = standard_form(lp)
    opt = DeviceMPC(d.A, d.b, d.c, d.l, d.u, c0=d.c0, objsense_min=d.objsense, device=0)
    return opt.optimize(), d


def compare(lp, obj_tol=1e-8, vec_tol=1e-6):
    dev, d = device_mpc(lp)
    ref, sh = solve_lp(lp, lambda A: HipBackend(A, device=0), algorithm="mpc")
    assert dev.status == ref.status
    assert abs(dev.niter - ref.niter) <= 1
    if ref.status == "Trm_Optimal":
        assert abs(dev.primal_objective - ref.primal_objective) <= obj_tol * (1 + abs(ref.primal_objective))
        assert abs(dev.dual_objective - ref.dual_objective) <= obj_tol * (1 + abs(ref.dual_objective))
        assert max(dev.rho) <= SQRT_EPS
        x = dev._get(0, dev.n)
        assert np.abs(x - ref.pt.x).max() <= vec_tol * max(1.0, np.abs(ref.pt.x).max())
    return dev, ref


@pytest.mark.gpu
def test_starting_point_and_iterates_side_by_side():
    """MPC.jl:353-410 on the device against the host-vector restatement, then three full iterations."""
    from test_ipm_harness import random_feasible_lp
    from tulip_jl_amd.mpc_device import DeviceMPC
    lp = random_feasible_lp(120, 300, 3, ineq=True)           # (m, n, seed)
    d = standard_form(lp)
    dev = DeviceMPC(d.A, d.b, d.c, d.l, d.u, c0=d.c0, device=0)
    ref = MPC(d, HipBackend(d.A, device=0))
    ref.compute_starting_point(); dev.compute_starting_point()
    pt = ref.pt
    for what, v in ((0, pt.x), (1, pt.xl), (2, pt.xu), (3, pt.zl), (4, pt.zu)):
        g = dev._get(what, dev.n)
        assert np.abs(g - v).max() <= 1e-10 * max(1.0, np.abs(v).max()), what
    assert np.abs(dev._get(5, dev.m) - pt.y).max() <= 1e-10 * max(1.0, np.abs(pt.y).max())
    assert abs(dev.mu - pt.mu) <= 1e-12 * pt.mu
    for it in range(3):
        ref.compute_residuals(); ref._update_mu(); dev.compute_residuals()
        for a, b in ((dev.rp_nrm, ref.rp_nrm), (dev.rd_nrm, ref.rd_nrm), (dev.rl_nrm, ref.rl_nrm), (dev.ru_nrm, ref.ru_nrm),
                     (dev.primal_objective, ref.primal_objective), (dev.dual_objective, ref.dual_objective), (dev.mu, pt.mu)):
            assert abs(a - b) <= 1e-8 * (1 + abs(b))
        ref.compute_step(); dev.compute_step()
        assert abs(dev.alpha_p - ref.alpha_p) <= 1e-7 and abs(dev.alpha_d - ref.alpha_d) <= 1e-7
        x = dev._get(0, dev.n); y = dev._get(5, dev.m)
        assert np.abs(x - pt.x).max() <= 1e-8 * max(1.0, np.abs(pt.x).max())
        assert np.abs(y - pt.y).max() <= 1e-8 * max(1.0, np.abs(pt.y).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lpex_opt.mps", "lpex_freevars.mps", "lpex_inf.mps", "lpex_ubd.mps"])
def test_reference_examples(name):
    dev, ref = compare(read_free_mps(os.path.join(GOLDEN, name)))
    assert dev.status == {"lpex_opt.mps": "Trm_Optimal", "lpex_freevars.mps": "Trm_Optimal",
                          "lpex_inf.mps": "Trm_PrimalInfeasible", "lpex_ubd.mps": "Trm_DualInfeasible"}[name]
    if name == "lpex_opt.mps":
        assert abs(dev.primal_objective - 1.5) <= 100 * SQRT_EPS


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_random_feasible_lps(seed):
    from test_ipm_harness import random_feasible_lp
    compare(random_feasible_lp(150 + 40 * seed, 400, 3 + seed, ineq=bool(seed)))


@pytest.mark.gpu
def test_netlib_class_lp_full_run():
    lp = read_free_mps(os.path.join(GOLDEN, "stair25.mps"))
    dev, ref = compare(lp, obj_tol=1e-7)
    assert dev.status == "Trm_Optimal"
    assert dev.timers["n_solve"] >= 2 * dev.niter + 2 and dev.timers["n_update"] == dev.niter + 1 + dev.timers["n_bump"]
