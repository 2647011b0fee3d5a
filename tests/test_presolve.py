"""Front end: presolve rules, scaling, postsolve and the Model flow.

Known answers come from the reference's own tests (test/Presolve/empty_column.jl, empty_row.jl, fixed_variable.jl),
from its example scripts (examples/*.jl run with the default Presolve level 1), and from HiGHS on generated LPs;
the interior-point method under the front end is the harness HSD over the CPU oracle backend here (no GPU), and
DeviceHSD on the MI355X in the `gpu` test.
"""
import itertools
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linprog

import tulip_jl_amd  # noqa: F401
from tulip_jl_amd.model import InnerResult, Model
from tulip_jl_amd.presolve import (DominatedColumn, DominatedRow, EmptyColumn, EmptyRow, FixedVariable, ForcingRow,
                                   FreeColumnSingleton, Presolve, RowSingleton)
from tulip_jl_amd.problem import LP, read_free_mps, standard_form
from ipm_harness import HSD, MPC, OracleBackend

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INF = float("inf")
TOL = 100 * float(np.sqrt(np.finfo(float).eps))          # examples/*.jl: atol = rtol = 100 sqrt(eps)


def cpu_ipm(algorithm="hsd"):
    def run(lp):
        d = standard_form(lp)
        ipm = (HSD if algorithm == "hsd" else MPC)(d, OracleBackend(d.A), None).optimize()
        pt = ipm.pt
        return InnerResult(ipm.status, ipm.primal_status, ipm.dual_status, pt.x, pt.y, pt.zl, pt.zu, pt.tau,
                           ipm.primal_objective, ipm.dual_objective, ipm.niter)
    return run


def one_var(l, u, c):
    return LP(sp.csc_matrix((0, 1)), [c], 0.0, [], [], [l], [u])


# ---- reference KATs -------------------------------------------------------------------------------
@pytest.mark.parametrize("l,u,c", list(itertools.product([-INF, -1.0], [1.0, INF], [-1.0, 0.0, 1.0])))
def test_empty_column_table(l, u, c):                     # test/Presolve/empty_column.jl:3-74
    ps = Presolve(one_var(l, u, c))
    ps._drop_empty_column(0)
    if c > 0 and not np.isfinite(l):
        assert ps.status == "Trm_DualInfeasible" and ps.colflag[0] and ps.ncol == 1
        sol = ps.solution
        assert sol.primal_status == "Sln_InfeasibilityCertificate" and (sol.m, sol.n) == (0, 1) and sol.x[0] < 0
    elif c < 0 and not np.isfinite(u):
        assert ps.status == "Trm_DualInfeasible" and ps.colflag[0] and ps.ncol == 1
        assert ps.solution.primal_status == "Sln_InfeasibilityCertificate" and ps.solution.x[0] > 0
    else:
        assert ps.status == "Trm_Unknown" and not ps.colflag[0] and ps.ncol == 0 and ps.updated
        assert len(ps.ops) == 1 and isinstance(ps.ops[0], EmptyColumn) and ps.ops[0].j == 0
        # the table of the reference's test: value the variable takes
        want = {(-INF, 1.0): {-1.0: 1.0, 0.0: 1.0}, (-INF, INF): {0.0: 0.0}, (-1.0, 1.0): {1.0: -1.0, -1.0: 1.0, 0.0: -1.0},
                (-1.0, INF): {1.0: -1.0, 0.0: -1.0}}[(l, u)][c]
        assert ps.ops[0].x == want


def test_fixed_variable_with_explicit_zeros():            # test/Presolve/fixed_variable.jl:4-40
    A = sp.csc_matrix((np.array([11.0, 0, 31, 0, 21, 0]), np.array([0, 1, 2, 0, 1, 2]), np.array([0, 3, 6])), shape=(3, 2))
    lp = LP(A, [1.0, 2.0], 0.0, np.zeros(3), np.ones(3), np.ones(2), np.ones(2))
    ps = Presolve(lp)
    assert ps.nzrow.tolist() == [1, 1, 1] and ps.nzcol.tolist() == [2, 1]
    ps._drop_fixed_variable(0)
    assert ps.colflag.tolist() == [False, True] and ps.obj0 == 1.0 and ps.nzrow.tolist() == [0, 1, 0]
    ps._drop_fixed_variable(1)
    assert ps.colflag.tolist() == [False, False] and ps.obj0 == 3.0 and ps.nzrow.tolist() == [0, 0, 0]


def test_empty_rows():                                    # test/Presolve/empty_row.jl:1-61
    lp = LP(sp.csc_matrix((2, 3)), np.ones(3), 0.0, [-1.0, 1.0], [1.0, 2.0], np.zeros(3), np.full(3, INF))
    ps = Presolve(lp)
    assert not ps.updated and ps.nzrow.tolist() == [0, 0]
    ps._drop_empty_row(0)
    assert ps.updated and ps.status == "Trm_Unknown" and ps.nrow == 1 and not ps.rowflag[0] and ps.rowflag[1]
    assert len(ps.ops) == 1 and isinstance(ps.ops[0], EmptyRow) and ps.ops[0].i == 0 and ps.ops[0].y == 0.0
    ps._drop_empty_row(1)                                 # 1 <= 0 <= 2: infeasible
    assert ps.status == "Trm_PrimalInfeasible" and ps.nrow == 1 and ps.rowflag[1] and len(ps.ops) == 1
    sol = ps.solution
    assert sol.dual_status == "Sln_InfeasibilityCertificate" and sol.z_primal == sol.z_dual == INF and sol.y_lower[0] > 0


@pytest.mark.parametrize("lo,up,which", [(1.0, 2.0, "y_lower"), (-2.0, -1.0, "y_upper")])
def test_empty_row_infeasible_single(lo, up, which):      # test/Presolve/empty_row.jl:64-140
    lp = LP(sp.csc_matrix((1, 1)), [1.0], 0.0, [lo], [up], [0.0], [INF])
    ps = Presolve(lp)
    ps._drop_empty_row(0)
    assert ps.status == "Trm_PrimalInfeasible" and ps.nrow == 1 and ps.rowflag[0] and not ps.ops
    assert getattr(ps.solution, which)[0] > 0 and ps.solution.z_primal == INF


# ---- rule by rule on small hand-made LPs -----------------------------------------------------------
def kkt_check(lp, sol, tol=1e-6):
    """Optimality conditions of  min c'x, lr <= Ax <= ur, lc <= x <= uc  in the original space."""
    sgn = 1.0 if lp.objsense_min else -1.0
    x, y, s = sol.x, sol.y_lower - sol.y_upper, sol.s_lower - sol.s_upper
    ax = lp.A @ x
    sc = 1 + max(np.abs(x).max(initial=0), np.abs(lp.obj).max(initial=0))
    assert (ax >= lp.lcon - tol * sc).all() and (ax <= lp.ucon + tol * sc).all(), "row bounds"
    assert (x >= lp.lvar - tol * sc).all() and (x <= lp.uvar + tol * sc).all(), "column bounds"
    assert np.abs(lp.A.T @ y + s - sgn * lp.obj).max(initial=0) <= tol * sc, "dual feasibility A'y + s = c"
    assert (sol.y_lower >= -tol).all() and (sol.y_upper >= -tol).all() and (sol.s_lower >= -tol).all() and (sol.s_upper >= -tol).all()
    # duals only on finite, active bounds
    with np.errstate(invalid="ignore"):
        assert np.abs(np.where(np.isfinite(lp.lcon), sol.y_lower * (ax - lp.lcon), sol.y_lower)).max(initial=0) <= tol * sc * sc
        assert np.abs(np.where(np.isfinite(lp.ucon), sol.y_upper * (lp.ucon - ax), sol.y_upper)).max(initial=0) <= tol * sc * sc
        assert np.abs(np.where(np.isfinite(lp.lvar), sol.s_lower * (x - lp.lvar), sol.s_lower)).max(initial=0) <= tol * sc * sc
        assert np.abs(np.where(np.isfinite(lp.uvar), sol.s_upper * (lp.uvar - x), sol.s_upper)).max(initial=0) <= tol * sc * sc


def highs(lp):
    sgn = 1.0 if lp.objsense_min else -1.0
    A = sp.csr_matrix(lp.A)
    eq = lp.lcon == lp.ucon
    ub_rows = np.isfinite(lp.ucon) & ~eq
    lb_rows = np.isfinite(lp.lcon) & ~eq
    Aub = sp.vstack([A[ub_rows], -A[lb_rows]]) if (ub_rows.any() or lb_rows.any()) else None
    bub = np.concatenate([lp.ucon[ub_rows], -lp.lcon[lb_rows]]) if Aub is not None else None
    r = linprog(sgn * lp.obj, A_ub=Aub, b_ub=bub, A_eq=A[eq] if eq.any() else None, b_eq=lp.lcon[eq] if eq.any() else None,
                bounds=[(None if l == -INF else l, None if u == INF else u) for l, u in zip(lp.lvar, lp.uvar)], method="highs")
    return r


def solve_and_check(lp, expect_ops=(), algorithm="hsd"):
    m = Model(lp).optimize(ipm=cpu_ipm(algorithm))
    kinds = {type(op) for op in m.presolve.ops}
    for k in expect_ops:
        assert k in kinds, f"{k.__name__} not applied (applied: {[t.__name__ for t in kinds]})"
    r = highs(lp)
    assert r.status == 0
    sgn = 1.0 if lp.objsense_min else -1.0
    assert m.status == "Trm_Optimal"
    assert abs(m.objective_value() - (sgn * r.fun + lp.obj0)) <= 1e-6 * (1 + abs(r.fun))
    assert abs(m.dual_objective_value() - m.objective_value()) <= 1e-6 * (1 + abs(r.fun))
    kkt_check(lp, m.solution)
    np.testing.assert_allclose(m.solution.Ax, lp.A @ m.solution.x)
    return m


def test_row_singleton_and_fixed_variable():
    # row 0: 2 x0 = 4 (singleton -> x0 fixed at 2), row 1: x0 + x1 + x2 >= 3, row 2: x1 - x2 <= 1
    A = sp.csc_matrix(np.array([[2.0, 0, 0], [1, 1, 1], [0, 1, -1]]))
    lp = LP(A, [1.0, 2.0, 3.0], 0.5, [4.0, 3.0, -INF], [4.0, INF, 1.0], [0.0, 0, 0], [10.0, INF, INF])
    m = solve_and_check(lp, (RowSingleton, FixedVariable))
    assert abs(m.solution.x[0] - 2.0) <= 1e-9


def test_row_singleton_negative_coefficient_duals():
    # -x0 >= -3  (i.e. x0 <= 3) with cost -1 on x0: the bound from the row is active, its dual must come back
    A = sp.csc_matrix(np.array([[-1.0, 0], [1.0, 1.0]]))
    lp = LP(A, [-1.0, 1.0], 0.0, [-3.0, 1.0], [INF, INF], [0.0, 0.0], [INF, INF])
    m = solve_and_check(lp, (RowSingleton,))
    assert abs(m.solution.x[0] - 3.0) <= 1e-6
    assert m.solution.y_lower[0] > 0.5                    # y_0 = 1: c_0 = -1 = a_00 y_0


def test_forcing_row():
    # x0 + x1 <= 0 with x >= 0 forces x0 = x1 = 0
    A = sp.csc_matrix(np.array([[1.0, 1.0, 0.0], [1.0, 2.0, 1.0], [0.0, 1.0, 1.0]]))
    lp = LP(A, [-1.0, -1.0, 1.0], 0.0, [-INF, 2.0, -INF], [0.0, INF, 5.0], [0.0, 0, 0], [INF, INF, INF])
    m = solve_and_check(lp, (ForcingRow,))
    assert abs(m.solution.x[0]) <= 1e-9 and abs(m.solution.x[1]) <= 1e-9


def test_forcing_row_reference_multiplier_option():
    """PresolveOptions.ReferenceForcingRowDual reproduces forcing_row.jl:196 as written (maximum for the record the
    reference stores in the minimal-activity case): same primal solution, reduced costs of the wrong sign."""
    from tulip_jl_amd.presolve import PresolveOptions
    A = sp.csc_matrix(np.array([[1.0, 1.0, 0.0], [1.0, 2.0, 1.0], [0.0, 1.0, 1.0]]))
    lp = LP(A, [-1.0, -1.0, 1.0], 0.0, [-INF, 2.0, -INF], [0.0, INF, 5.0], [0.0, 0, 0], [INF, INF, INF])
    good = Model(lp).optimize(ipm=cpu_ipm())
    lit = Model(lp)
    lit.presolve_options = PresolveOptions(ReferenceForcingRowDual=True)
    lit.optimize(ipm=cpu_ipm())
    np.testing.assert_allclose(lit.solution.x, good.solution.x, atol=1e-9)
    assert lit.solution.s_upper.max() > 0.5 and good.solution.s_upper.max() <= 1e-9   # x >= 0 has no upper bound to price


def test_dominated_row():
    # x0 + x1 <= 10 can never bind when 0 <= x <= 2
    A = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, -1.0], [1.0, 2.0]]))
    lp = LP(A, [-1.0, -2.0], 0.0, [-INF, -1.0, -INF], [10.0, 1.0, 5.0], [0.0, 0.0], [2.0, 2.0])
    m = solve_and_check(lp, (DominatedRow,))
    assert m.solution.y_lower[0] == 0.0 and m.solution.y_upper[0] == 0.0


def test_free_column_singleton():
    # x2 is free and appears only in row 0 (an equality): it is substituted out
    A = sp.csc_matrix(np.array([[1.0, 1.0, 2.0], [1.0, -1.0, 0.0], [1.0, 1.0, 0.0]]))
    lp = LP(A, [1.0, 1.0, 0.5], 0.0, [4.0, -1.0, 1.0], [4.0, 1.0, INF], [0.0, 0.0, -INF], [3.0, 3.0, INF])
    solve_and_check(lp, (FreeColumnSingleton,))


def test_free_column_singleton_with_a_multiplier_of_the_wrong_sign_is_unbounded():
    """x2 is free, appears only in the one-sided row  x0 + 2 x2 >= 1  and has cost -1: its dual constraint 2 y = -1 needs
    y < 0 on a row that has no upper bound -> dual infeasible.  free_column_singleton.jl:79 adds y * Inf to the objective
    constant instead (the reference option reproduces that: the interior-point run then cannot terminate properly)."""
    from tulip_jl_amd.presolve import Presolve, PresolveOptions
    A = sp.csc_matrix(np.array([[1.0, 0.0, 2.0], [1.0, 1.0, 0.0]]))
    lp = LP(A, [1.0, 1.0, -1.0], 0.0, [1.0, 0.5], [INF, 2.0], [0.0, 0.0, -INF], [3.0, 3.0, INF])
    assert highs(lp).status == 3                          # HiGHS: unbounded
    m = Model(lp).optimize(ipm=cpu_ipm())
    assert m.status == "Trm_DualInfeasible" and m.inner is None
    d = m.solution.x
    assert m.solution.primal_status == "Sln_InfeasibilityCertificate" and d[2] > 0
    assert float(lp.obj @ d) < 0 and ((lp.A @ d)[0] >= 0) and abs((lp.A @ d)[1]) <= 1e-12 and d[0] == 0 and d[1] == 0
    ps = Presolve(lp, PresolveOptions(ReferenceUnboundedSingleton=True))
    assert ps.run() == "Trm_Unknown" and ps.obj0 == -INF  # as the reference: objective constant -Inf, no verdict


def test_zero_cost_free_column_singleton_in_a_one_sided_row():
    """min x0 + x1  s.t.  x0 + x1 + s <= 5 (s free, cost 0),  x0 - x1 >= 1: the singleton's multiplier is y = 0 and the row
    bound it would price is -Inf; nothing may be added to the objective constant (0 * Inf = NaN) and postsolve must place
    s on a FINITE row activity (round-2 advisor finding: x = [1, 0, -inf], objective NaN, status optimal)."""
    A = sp.csc_matrix(np.array([[1.0, 1.0, 1.0], [1.0, -1.0, 0.0]]))
    lp = LP(A, [1.0, 1.0, 0.0], 0.0, [-INF, 1.0], [5.0, INF], [0.0, 0.0, -INF], [INF, INF, INF])
    m = solve_and_check(lp, (FreeColumnSingleton,))
    assert np.isfinite(m.presolve.obj0) and np.all(np.isfinite(m.solution.x))
    assert abs(m.objective_value() - 1.0) <= 1e-6
    # a free row keeps the activity 0
    lp2 = LP(A, [1.0, 1.0, 0.0], 0.0, [-INF, 1.0], [INF, INF], [0.0, 0.0, -INF], [INF, INF, INF])
    m2 = Model(lp2).optimize(ipm=cpu_ipm())
    assert m2.status == "Trm_Optimal" and np.all(np.isfinite(m2.solution.x)) and abs(m2.objective_value() - 1.0) <= 1e-6


def test_forcing_row_ignores_explicit_zeros():
    """An explicit zero in a forcing row whose column has an infinite bound: 0 * Inf must not poison the other rows' bounds
    (round-2 advisor finding: standard_form raised 'Invalid bounds for row 0: [-inf, nan]')."""
    # row 0: x0 + x1 + 0 * x2 <= 0 with x0, x1 >= 0 (forcing), x2 free with an explicit zero in row 0
    A = sp.csc_matrix((np.array([1.0, 1.0, 1.0, 2.0, 1.0, 0.0, 1.0, 1.0]), np.array([0, 1, 0, 1, 2, 0, 1, 2]), np.array([0, 2, 5, 8])), shape=(3, 3))
    lp = LP(A, [-1.0, -1.0, 1.0], 0.0, [-INF, 2.0, -INF], [0.0, INF, 5.0], [0.0, 0.0, -INF], [INF, INF, INF])
    m = solve_and_check(lp, (ForcingRow,))
    assert abs(m.solution.x[0]) <= 1e-9 and abs(m.solution.x[1]) <= 1e-9 and np.all(np.isfinite(m.solution.x))


def test_dominated_column():
    # x2 >= 0 has cost +5 and only helps "<=" rows' slack the wrong way: reduced cost always positive -> lower bound
    A = sp.csc_matrix(np.array([[1.0, 1.0, 1.0], [1.0, -1.0, 2.0]]))
    lp = LP(A, [-1.0, -1.0, 5.0], 0.0, [-INF, -INF], [4.0, 2.0], [0.0, 0.0, 0.0], [INF, INF, INF])
    m = solve_and_check(lp, (DominatedColumn,))
    assert m.solution.x[2] == 0.0


def test_maximisation_and_objective_constant():
    A = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, 3.0], [1.0, 0.0]]))
    lp = LP(A, [2.0, 3.0], 7.0, [-INF, -INF, -INF], [4.0, 6.0, 3.0], [0.0, 0.0], [INF, INF], objsense_min=False)
    m = solve_and_check(lp, (RowSingleton,))
    assert abs(m.objective_value() - (2 * 3 + 3 * 1 + 7)) <= 1e-6


def test_presolve_decides_infeasible_and_unbounded():
    lp = LP(sp.csc_matrix(np.array([[1.0, 1.0]])), [1.0, 1.0], 0.0, [3.0], [2.0], [0.0, 0.0], [INF, INF])   # l > u
    m = Model(lp).optimize(ipm=None if False else cpu_ipm())
    assert m.status == "Trm_PrimalInfeasible" and m.inner is None and m.solution.dual_status == "Sln_InfeasibilityCertificate"
    assert m.solution.y_lower[0] == 1.0 and m.solution.y_upper[0] == 1.0
    lp = LP(sp.csc_matrix(np.array([[1.0, 0.0]])), [1.0, -1.0], 0.0, [0.0], [2.0], [0.0, 0.0], [INF, INF])  # empty column, c < 0, u = inf
    m = Model(lp).optimize(ipm=cpu_ipm())
    assert m.status == "Trm_DualInfeasible" and m.solution.primal_status == "Sln_InfeasibilityCertificate" and m.solution.x[1] > 0


def test_presolve_solves_everything():
    # every row is a singleton: presolve ends with an empty problem and declares optimality (Presolve.jl:409-421)
    lp = LP(sp.csc_matrix(np.array([[2.0, 0.0], [0.0, 1.0]])), [1.0, 1.0], 3.0, [2.0, 1.0], [2.0, 1.0], [0.0, 0.0], [INF, INF])
    m = Model(lp).optimize(ipm=cpu_ipm())
    assert m.status == "Trm_Optimal" and m.inner is None
    np.testing.assert_allclose(m.solution.x, [1.0, 1.0])
    assert m.objective_value() == 5.0 and m.solution.z_primal == 5.0
    kkt_check(lp, m.solution)


# ---- scaling --------------------------------------------------------------------------------------
def test_lpex_opt_is_only_rescaled():
    """presolve removes nothing from lpex_opt.mps but rescales A by
    sqrt(||row||_2) sqrt(||col||_2) (Presolve.jl:256-295): all four norms are sqrt(2), so A / sqrt(2)."""
    lp = read_free_mps(os.path.join(GOLDEN, "lpex_opt.mps"))
    ps = Presolve(lp)
    assert ps.run() == "Trm_Unknown" and not ps.ops and (ps.nrow, ps.ncol) == (2, 2)
    red = ps.reduced_problem()
    r2 = 2 ** 0.25
    np.testing.assert_allclose(ps.row_scaling, [r2, r2]); np.testing.assert_allclose(ps.col_scaling, [r2, r2])
    np.testing.assert_allclose(red.A.toarray(), np.array([[1, 1], [1, -1]]) / np.sqrt(2.0))
    np.testing.assert_allclose(red.obj, np.array([1, 2]) / r2)
    np.testing.assert_allclose(red.uvar, np.array([1, 1]) * r2)
    np.testing.assert_allclose(red.lcon, np.array([1, 0]) / r2)


def test_scaling_identity_on_random_matrix():
    rng = np.random.default_rng(5)
    A = sp.random(30, 50, density=0.2, random_state=5, format="csc") * 100
    lp = LP(A, rng.normal(size=50), 0.0, np.full(30, -INF), rng.uniform(1, 2, 30) * 50, np.zeros(50), np.full(50, 10.0))
    ps = Presolve(lp)
    ps.run()
    red = ps.reduced_problem()
    sub = sp.csc_matrix(lp.A)[ps.old_con_idx][:, ps.old_var_idx]
    sub.eliminate_zeros()
    back = sp.diags(ps.row_scaling) @ red.A @ sp.diags(ps.col_scaling)
    assert abs(back - sub).max() <= 1e-12 * abs(sub).max()
    rn = np.sqrt(np.asarray(sub.multiply(sub).sum(axis=1)).ravel()); cn = np.sqrt(np.asarray(sub.multiply(sub).sum(axis=0)).ravel())
    np.testing.assert_allclose(ps.row_scaling, np.sqrt(np.where(rn > 0, rn, 1.0)))
    np.testing.assert_allclose(ps.col_scaling, np.sqrt(np.where(cn > 0, cn, 1.0)))


# ---- end to end: the reference's examples with its default presolve level --------------------------
def test_example_optimal():                               # examples/optimal.jl:37-62
    m = Model.load(os.path.join(GOLDEN, "lpex_opt.mps")).optimize(ipm=cpu_ipm())
    assert m.status == "Trm_Optimal" and abs(m.objective_value() - 1.5) <= TOL
    s = m.solution
    assert s.primal_status == "Sln_Optimal" and s.dual_status == "Sln_Optimal"
    np.testing.assert_allclose(s.x, [0.5, 0.5], atol=TOL); np.testing.assert_allclose(s.Ax, [1.0, 0.0], atol=TOL)
    np.testing.assert_allclose(s.y_lower - s.y_upper, [1.5, -0.5], atol=TOL)
    np.testing.assert_allclose(s.s_lower - s.s_upper, [0.0, 0.0], atol=TOL)


def test_example_freevars():                              # examples/freevars.jl:35-57
    m = Model.load(os.path.join(GOLDEN, "lpex_freevars.mps")).optimize(ipm=cpu_ipm())
    assert m.status == "Trm_Optimal" and abs(m.objective_value()) <= TOL
    x = m.solution.x
    assert 2 * x[0] + x[1] >= 2 - TOL and x[0] + 2 * x[1] >= 2 - TOL and x[0] + x[1] + x[2] >= -TOL


def test_example_infeasible():                            # examples/infeasible.jl:36-53
    m = Model.load(os.path.join(GOLDEN, "lpex_inf.mps")).optimize(ipm=cpu_ipm())
    assert m.status == "Trm_PrimalInfeasible"
    s = m.solution
    assert s.primal_status == "Sln_Unknown" and s.dual_status == "Sln_InfeasibilityCertificate"
    y = s.y_lower - s.y_upper
    sl, su = s.s_lower, s.s_upper
    lp = m.lp
    # Farkas: A'y + s = 0, and the dual ray objective is positive
    np.testing.assert_allclose(lp.A.T @ y + (sl - su), 0.0, atol=TOL)
    assert m.dual_objective_value() >= TOL


def test_example_unbounded():                             # examples/unbounded.jl:34-53
    m = Model.load(os.path.join(GOLDEN, "lpex_ubd.mps")).optimize(ipm=cpu_ipm())
    assert m.status == "Trm_DualInfeasible"
    s = m.solution
    assert s.primal_status == "Sln_InfeasibilityCertificate" and s.dual_status == "Sln_Unknown"
    assert s.x[0] >= -TOL and s.x[1] >= -TOL and abs(s.Ax[0]) <= TOL and m.objective_value() <= -TOL


@pytest.mark.parametrize("name", ["stair25.mps", "bump.mps"])
@pytest.mark.parametrize("algorithm", ["hsd", "mpc"])
def test_generated_netlib_class_lps_with_presolve(name, algorithm):
    """The generated stand-ins of the Netlib configurations (tests/lp_generators.py) through presolve + scaling:
    same optimum as HiGHS, optimality conditions in the ORIGINAL space."""
    lp = read_free_mps(os.path.join(GOLDEN, name))
    m = Model(lp).optimize(ipm=cpu_ipm(algorithm))
    r = highs(lp)
    assert r.status == 0 and m.status == "Trm_Optimal"
    assert abs(m.objective_value() - (r.fun + lp.obj0)) <= 1e-6 * (1 + abs(r.fun))
    kkt_check(lp, m.solution, tol=1e-5)
    assert m.presolve.nrow <= lp.A.shape[0] and m.presolve.ncol <= lp.A.shape[1]


@pytest.mark.parametrize("seed", range(16))
def test_random_lps_with_reducible_structure(seed):
    """Random feasible LPs salted with fixed variables, singleton rows, empty rows / columns, free singleton columns
    and redundant rows: presolve + postsolve must return an optimal primal-dual pair of the original problem."""
    rng = np.random.default_rng(100 + seed)
    m_, n_ = 25, 40
    A = np.where(rng.random((m_, n_)) < 0.15, rng.integers(1, 5, (m_, n_)) * rng.choice([-1.0, 1.0], (m_, n_)), 0.0)
    xs = rng.uniform(0, 2, n_)
    lvar, uvar = np.zeros(n_), np.full(n_, 4.0)
    for j in rng.choice(n_, 4, replace=False):             # fixed variables
        lvar[j] = uvar[j] = xs[j]
    jfree = int(rng.integers(12, n_))                      # a free singleton column
    A[:, jfree] = 0; A[0, jfree] = 2.0; lvar[jfree], uvar[jfree] = -INF, INF
    A[3, :] = 0; A[3, 5] = 3.0                             # singleton row
    A[7, :] = 0                                            # empty row
    A[:, 11] = 0                                           # empty column
    A = sp.csc_matrix(A)
    ax = A @ xs
    lcon, ucon = ax - rng.uniform(0, 1, m_), ax + rng.uniform(0, 1, m_)
    eq = rng.random(m_) < 0.3
    lcon[eq] = ucon[eq] = ax[eq]
    lcon[7], ucon[7] = -1.0, 1.0
    ucon[rng.random(m_) < 0.2] = INF
    lcon[9], ucon[9] = -1e6, 1e6                           # redundant row
    obj = rng.normal(size=n_)
    obj[11] = abs(obj[11])
    lp = LP(A, obj if seed % 3 else -obj, 1.25, lcon, ucon, lvar, uvar, objsense_min=bool(seed % 3))   # every third one maximises
    r = highs(lp)
    if r.status != 0:                                      # HiGHS: 2 infeasible, 3 unbounded -- the front end must agree
        mod = Model(lp).optimize(ipm=cpu_ipm())
        assert r.status in (2, 3), r.message
        assert mod.status == {2: "Trm_PrimalInfeasible", 3: "Trm_DualInfeasible"}[r.status]
        return
    mod = solve_and_check(lp, (FixedVariable, EmptyRow, EmptyColumn))
    assert mod.presolve.nrow < m_ and mod.presolve.ncol < n_


@pytest.mark.gpu
@pytest.mark.parametrize("name,obj", [("lpex_opt.mps", 1.5), ("lpex_freevars.mps", 0.0)])
def test_model_on_device(name, obj):
    """The product flow: MPS -> presolve + scaling -> DeviceHSD on the MI355X -> postsolve."""
    m = Model.load(os.path.join(GOLDEN, name)).optimize()
    assert m.status == "Trm_Optimal" and abs(m.objective_value() - obj) <= TOL
    kkt_check(m.lp, m.solution, tol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", ["hsd", "mpc"])
def test_model_on_device_netlib_class(algorithm):
    lp = read_free_mps(os.path.join(GOLDEN, "stair25.mps"))
    m = Model(lp, algorithm=algorithm).optimize()
    ref = Model(lp).optimize(ipm=cpu_ipm(algorithm))
    r = highs(lp)
    assert m.status == "Trm_Optimal" and abs(m.objective_value() - r.fun) <= 1e-6 * (1 + abs(r.fun))
    assert abs(m.objective_value() - ref.objective_value()) <= 1e-7 * (1 + abs(r.fun))
    assert abs(m.inner.niter - ref.inner.niter) <= 1
    kkt_check(lp, m.solution, tol=1e-5)
