"""Presolve, scaling and postsolve of the LP front end.

Restates the behaviour of /root/reference/src/Presolve/ so that the interior-point method -- and
therefore KKT.setup -- sees the same reduced, rescaled constraint matrix Tulip would hand over:

    min  c'x + c0      lr <= A x <= ur,   lc <= x <= uc          (maximisation is negated on entry)

  rules           reference                              here
  --------------  -------------------------------------  ---------------------------------
  bound checks    Presolve.jl:463-527                    Presolve._check_bounds
  empty row       empty_row.jl:9-68                      Presolve._drop_empty_row
  empty column    empty_column.jl:7-97                   Presolve._drop_empty_column
  fixed variable  fixed_variable.jl:8-55                 Presolve._drop_fixed_variable
  row singleton   row_singleton.jl:10-74                 Presolve._drop_row_singleton
  forcing / dominated row  forcing_row.jl:14-172         Presolve._forcing_row
  (implied) free column singleton  free_column_singleton.jl:11-109   Presolve._free_column_singleton
  dominated column (+ dual bounds from singletons)  Presolve.jl:640-707, dominated_column.jl:8-139
                                                         Presolve._dominated_columns
  driver          Presolve.jl:374-452                    Presolve.run
  reduced problem + scaling by sqrt(||row||_2) sqrt(||col||_2)   Presolve.jl:177-305   Presolve.reduced_problem
  postsolve       Presolve.jl:320-365 + the rule files   Presolve.postsolve

The pass order, the tolerances (sqrt(eps) for empty rows / columns, 100 sqrt(eps) for dominated columns), the
recorded transformations and their reversal follow the reference, quirks included and marked `quirk:` (they decide
which rows and columns survive, i.e. the matrix the KKT backend factorises).  Rows and columns are never
physically deleted during the passes: two activity masks over the ORIGINAL matrix, as in the reference.

Data layout: the original matrix once as CSR and once as CSC (explicit zeros kept); every rule is a loop over one
row or one column of those.  Host-side Python: this runs once per model, before the first KKT.setup.
"""
import math
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from .problem import LP

INF = float("inf")
SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))

TRM_UNKNOWN = "Trm_Unknown"
TRM_OPTIMAL = "Trm_Optimal"
TRM_PRIMAL_INFEASIBLE = "Trm_PrimalInfeasible"
TRM_DUAL_INFEASIBLE = "Trm_DualInfeasible"
SLN_UNKNOWN = "Sln_Unknown"
SLN_OPTIMAL = "Sln_Optimal"
SLN_FEASIBLE = "Sln_FeasiblePoint"
SLN_RAY = "Sln_InfeasibilityCertificate"


def _pos(v):
    return v if v >= 0.0 else 0.0


def _neg(v):
    return -v if v <= 0.0 else 0.0


class Solution:
    """/root/reference/src/solution.jl: primal x, row activity Ax, duals split into non-negative parts
    (y = y_lower - y_upper for rows, s = s_lower - s_upper for columns)."""

    def __init__(self, m, n):
        self.resize(m, n)
        self.primal_status = self.dual_status = SLN_UNKNOWN
        self.is_primal_ray = self.is_dual_ray = False
        self.z_primal = self.z_dual = float("nan")

    def resize(self, m, n):
        self.m, self.n = m, n
        self.x = np.zeros(n); self.Ax = np.zeros(m)
        self.y_lower = np.zeros(m); self.y_upper = np.zeros(m)
        self.s_lower = np.zeros(n); self.s_upper = np.zeros(n)


@dataclass
class PresolveOptions:                      # Presolve.jl:1-6
    Level: int = 1
    TolerancePFeas: float = SQRT_EPS
    ToleranceDFeas: float = SQRT_EPS
    ReferenceForcingRowDual: bool = False   # see _undo_forcing_row
    ReferenceUnboundedSingleton: bool = False   # see _free_column_singleton


# recorded transformations (reversed in postsolve, last first)
@dataclass
class EmptyRow:
    i: int
    y: float


@dataclass
class EmptyColumn:
    j: int
    x: float
    s: float


@dataclass
class FixedVariable:
    j: int
    x: float
    c: float
    rows: np.ndarray
    vals: np.ndarray


@dataclass
class RowSingleton:
    i: int
    j: int
    aij: float
    force_lower: bool
    force_upper: bool


@dataclass
class DominatedRow:
    i: int


@dataclass
class ForcingRow:
    i: int
    at_lower: bool
    cols: list                   # column indices of the row's live entries
    vals: list                   # their coefficients
    col_rows: list = field(default_factory=list)   # per column: live rows / values at removal time
    col_vals: list = field(default_factory=list)
    xs: list = field(default_factory=list)
    cs: list = field(default_factory=list)
    literal: bool = False        # PresolveOptions.ReferenceForcingRowDual


@dataclass
class FreeColumnSingleton:
    i: int
    j: int
    l: float
    u: float
    aij: float
    y: float
    cols: list
    vals: list


@dataclass
class DominatedColumn:
    j: int
    x: float
    c: float
    rows: np.ndarray
    vals: np.ndarray


class Presolve:
    def __init__(self, lp, options=None):
        self.lp = lp
        self.opt = options or PresolveOptions()
        self.A_csc = sp.csc_matrix(lp.A)
        self.A_csr = sp.csr_matrix(lp.A)
        self.m0, self.n0 = self.A_csc.shape
        self.status = TRM_UNKNOWN
        self.updated = False
        self.solution = Solution(self.m0, self.n0)            # only meaningful when presolve decides the problem
        self.nrow, self.ncol = self.m0, self.n0
        self.rowflag = np.ones(self.m0, bool)
        self.colflag = np.ones(self.n0, bool)
        # non-zero counts (explicit zeros do not count, Presolve.jl:119-124)
        nz = self.A_csc.data != 0
        run = np.concatenate([[0], np.cumsum(nz)]).astype(np.int64)
        self.nzcol = run[self.A_csc.indptr[1:]] - run[self.A_csc.indptr[:-1]]
        self.nzrow = np.bincount(self.A_csc.indices[nz], minlength=self.m0).astype(np.int64)
        self.objsense_min = lp.objsense_min
        sgn = 1.0 if lp.objsense_min else -1.0               # maximisation: negated here, undone in reduced_problem
        self.obj = sgn * np.array(lp.obj, float)
        self.obj0 = sgn * float(lp.obj0)
        self.lrow = np.array(lp.lcon, float); self.urow = np.array(lp.ucon, float)
        self.lcol = np.array(lp.lvar, float); self.ucol = np.array(lp.uvar, float)
        # dual bounds (Presolve.jl:144-155)
        self.ly = np.where(self.urow == INF, 0.0, -INF); self.uy = np.where(self.lrow == -INF, 0.0, INF)
        self.ls = np.where(self.ucol == INF, 0.0, -INF); self.us = np.where(self.lcol == -INF, 0.0, INF)
        self.row_scaling = np.ones(self.m0); self.col_scaling = np.ones(self.n0)
        self.new_con_idx = self.new_var_idx = self.old_con_idx = self.old_var_idx = None
        self.row_singletons = []
        self.ops = []
        self.reduced = None

    # -- views of one row / one column of the original matrix -----------------------------------
    def _row(self, i):
        a, b = self.A_csr.indptr[i], self.A_csr.indptr[i + 1]
        return self.A_csr.indices[a:b], self.A_csr.data[a:b]

    def _col(self, j):
        a, b = self.A_csc.indptr[j], self.A_csc.indptr[j + 1]
        return self.A_csc.indices[a:b], self.A_csc.data[a:b]

    # -- Presolve.jl:431-461 ---------------------------------------------------------------------
    def _index_mapping(self):
        self.old_con_idx = np.flatnonzero(self.rowflag)
        self.old_var_idx = np.flatnonzero(self.colflag)
        self.new_con_idx = np.full(self.m0, -1, np.int64); self.new_con_idx[self.old_con_idx] = np.arange(self.nrow)
        self.new_var_idx = np.full(self.n0, -1, np.int64); self.new_var_idx[self.old_var_idx] = np.arange(self.ncol)

    def _decided(self, status, primal_ray=None, dual_ray=None):
        """Presolve proved infeasibility / unboundedness: status + certificate in the CURRENT reduced index space."""
        self.status = status
        self.updated = True
        self._index_mapping()
        sol = self.solution
        sol.resize(self.nrow, self.ncol)
        if status == TRM_PRIMAL_INFEASIBLE:                   # Farkas ray
            sol.primal_status, sol.dual_status = SLN_UNKNOWN, SLN_RAY
            sol.is_primal_ray, sol.is_dual_ray = False, True
            sol.z_primal = sol.z_dual = INF
            dual_ray(sol)
        else:                                                 # unbounded ray
            sol.primal_status, sol.dual_status = SLN_RAY, SLN_UNKNOWN
            sol.is_primal_ray, sol.is_dual_ray = True, False
            sol.z_primal = sol.z_dual = -INF
            primal_ray(sol)

    # -- Presolve.jl:463-527 ---------------------------------------------------------------------
    def _check_bounds(self):
        for i in np.flatnonzero(self.rowflag & (self.lrow > self.urow)):
            def ray(sol, i=i):
                sol.y_lower[self.new_con_idx[i]] = 1.0; sol.y_upper[self.new_con_idx[i]] = 1.0
            self._decided(TRM_PRIMAL_INFEASIBLE, dual_ray=ray)
            return
        for j in np.flatnonzero(self.colflag & (self.lcol > self.ucol)):
            def ray(sol, j=j):
                sol.s_lower[self.new_var_idx[j]] = 1.0; sol.s_upper[self.new_var_idx[j]] = 1.0
            self._decided(TRM_PRIMAL_INFEASIBLE, dual_ray=ray)
            return

    # -- empty_row.jl:9-68 -----------------------------------------------------------------------
    def _drop_empty_row(self, i):
        if not (self.rowflag[i] and self.nzrow[i] == 0):
            return
        lb, ub, eps = self.lrow[i], self.urow[i], self.opt.TolerancePFeas
        if ub < -eps:
            def ray(sol):
                sol.y_upper[self.new_con_idx[i]] = 1.0
            return self._decided(TRM_PRIMAL_INFEASIBLE, dual_ray=ray)
        if lb > eps:
            def ray(sol):
                sol.y_lower[self.new_con_idx[i]] = 1.0
            return self._decided(TRM_PRIMAL_INFEASIBLE, dual_ray=ray)
        self.ops.append(EmptyRow(i, 0.0))
        self.updated = True
        self.rowflag[i] = False
        self.nrow -= 1

    # -- empty_column.jl:7-97 --------------------------------------------------------------------
    def _drop_empty_column(self, j):
        if not (self.colflag[j] and self.nzcol[j] == 0):
            return
        lb, ub, cj, eps = self.lcol[j], self.ucol[j], self.obj[j], self.opt.ToleranceDFeas
        if cj > eps:
            if math.isfinite(lb):
                self.obj0 += lb * cj
                self.ops.append(EmptyColumn(j, lb, cj))
            else:
                def ray(sol):
                    sol.x[self.new_var_idx[j]] = -1.0
                return self._decided(TRM_DUAL_INFEASIBLE, primal_ray=ray)
        elif cj < -eps:
            if math.isfinite(ub):
                self.obj0 += ub * cj
                self.ops.append(EmptyColumn(j, ub, cj))
            else:
                def ray(sol):
                    sol.x[self.new_var_idx[j]] = 1.0
                return self._decided(TRM_DUAL_INFEASIBLE, primal_ray=ray)
        else:                                                 # any feasible value
            v = lb if math.isfinite(lb) else (ub if math.isfinite(ub) else 0.0)
            self.ops.append(EmptyColumn(j, v, 0.0))
        self.colflag[j] = False
        self.updated = True
        self.ncol -= 1

    # -- fixed_variable.jl:8-55 ------------------------------------------------------------------
    def _drop_fixed_variable(self, j):
        if not self.colflag[j]:
            return
        lb, ub = self.lcol[j], self.ucol[j]
        if lb != ub:
            return
        rows, vals = self._col(j)
        cj = self.obj[j]
        self.colflag[j] = False
        self.ncol -= 1
        self.updated = True
        live = self.rowflag[rows]
        self.ops.append(FixedVariable(j, lb, cj, rows[live].copy(), vals[live].copy()))
        self.obj0 += cj * lb
        for i, aij in zip(rows, vals):
            if not self.rowflag[i] or aij == 0.0:
                continue
            self.lrow[i] -= aij * lb
            self.urow[i] -= aij * lb
            self.nzrow[i] -= 1
            if self.nzrow[i] == 0:
                self._drop_empty_row(i)
            # quirk: fixed_variable.jl:39 compares the whole count VECTOR with 1, which is never true: a row
            # that becomes a singleton here is not queued (it is still caught by the forcing-row test later)

    # -- row_singleton.jl:10-74 ------------------------------------------------------------------
    def _drop_row_singleton(self, i):
        if not (self.rowflag[i] and self.nzrow[i] == 1):
            return
        cols, vals = self._row(i)
        nz, j, aij = 0, -1, 0.0
        for j_, a_ in zip(cols, vals):
            if self.colflag[j_] and a_ != 0.0:
                nz += 1
                if nz > 1:
                    break
                j, aij = j_, a_
        if nz > 1 or aij == 0.0:
            return
        if aij > 0.0:
            l, u = self.lrow[i] / aij, self.urow[i] / aij
        else:
            l, u = self.urow[i] / aij, self.lrow[i] / aij
        lb, ub = self.lcol[j], self.ucol[j]
        force_lower, force_upper = bool(l >= lb), bool(u <= ub)
        if force_lower:
            self.lcol[j] = l
        if force_upper:
            self.ucol[j] = u
        self.ops.append(RowSingleton(i, j, aij, force_lower, force_upper))
        self.rowflag[i] = False
        self.updated = True
        self.nrow -= 1
        self.nzcol[j] -= 1
        if self.lcol[j] == self.ucol[j]:
            self._drop_fixed_variable(j)

    def _drop_row_singletons(self):                          # Presolve.jl:585-592
        for i in self.row_singletons:
            self._drop_row_singleton(i)
        self.row_singletons = []

    # -- forcing_row.jl:14-172 -------------------------------------------------------------------
    def _forcing_row(self, i):
        if not self.rowflag[i] or self.nzrow[i] == 1:
            return
        cols, vals = self._row(i)
        lo = up = 0.0
        with np.errstate(invalid="ignore"):
            for j, aij in zip(cols, vals):
                if not self.colflag[j] or aij == 0.0:         # an explicit zero is not an entry of the row (0 * Inf = NaN otherwise)
                    continue
                if aij < 0.0:
                    lo += aij * self.ucol[j]; up += aij * self.lcol[j]
                else:
                    lo += aij * self.lcol[j]; up += aij * self.ucol[j]
                if not (math.isfinite(lo) or math.isfinite(up)):
                    break
        l, u = self.lrow[i], self.urow[i]
        if l <= lo <= up <= u:                                # dominated: never active
            self.rowflag[i] = False
            self.updated = True
            self.nrow -= 1
            self.ops.append(DominatedRow(i))
            for j, aij in zip(cols, vals):
                if self.colflag[j]:
                    self.nzcol[j] -= int(aij != 0.0)
            return
        if lo == u:
            at_lower = True          # naming of forcing_row.jl:51,102: the record says `true` when the minimal activity meets u
        elif up == l:
            at_lower = False
        else:
            return
        live = self.colflag[cols] & (vals != 0.0)             # explicit zeros neither fix their column nor enter the record
        op = ForcingRow(i, at_lower, [int(j) for j in cols[live]], [float(a) for a in vals[live]], literal=self.opt.ReferenceForcingRowDual)
        for j, aij in zip(cols, vals):
            if not self.colflag[j] or aij == 0.0:
                continue
            if at_lower:
                xj = self.lcol[j] if aij > 0 else self.ucol[j]
            else:
                xj = self.ucol[j] if aij > 0 else self.lcol[j]
            rws, cvs = self._col(j)
            keep_r, keep_v = [], []
            for k, akj in zip(rws, cvs):
                if not self.rowflag[k]:
                    continue
                keep_r.append(int(k)); keep_v.append(float(akj))
                self.nzrow[k] -= 1                            # (explicit zeros included, as in the reference)
                self.lrow[k] -= akj * xj
                self.urow[k] -= akj * xj
                if self.nzrow[k] == 1:
                    self.row_singletons.append(int(k))
            op.col_rows.append(keep_r); op.col_vals.append(keep_v)
            op.xs.append(float(xj)); op.cs.append(float(self.obj[j]))
            self.colflag[j] = False
            self.ncol -= 1
        self.ops.append(op)
        self.rowflag[i] = False
        self.nrow -= 1
        self.updated = True

    # -- free_column_singleton.jl:11-109 ---------------------------------------------------------
    def _free_column_singleton(self, j):
        if not (self.colflag[j] and self.nzcol[j] == 1):
            return
        rows, vals = self._col(j)
        nz, i, aij = 0, -1, 0.0
        for i_, a_ in zip(rows, vals):
            if self.rowflag[i_]:
                nz += int(a_ != 0.0)
                if nz > 1:
                    break
                i, aij = int(i_), float(a_)                   # quirk: the LAST live entry seen, explicit zeros included
        if nz != 1:
            raise RuntimeError(f"Expected singletons but column {j} has {nz} non-zeros")
        if aij == 0.0 or i < 0:
            return
        cols, rvals = self._row(i)
        lr, ur = self.lrow[i], self.urow[i]
        l, u = self.lcol[j], self.ucol[j]
        if math.isfinite(l) or math.isfinite(u):              # not free: implied bounds from the row
            lo, up = (lr, ur) if aij > 0 else (ur, lr)
            with np.errstate(invalid="ignore"):
                for k, aik in zip(cols, rvals):
                    if not self.colflag[k] or k == j:
                        continue
                    if (aik > 0) == (aij > 0):
                        lo -= aik * self.ucol[k]; up -= aik * self.lcol[k]
                    else:
                        lo -= aik * self.lcol[k]; up -= aik * self.ucol[k]
            lo /= aij; up /= aij
            if not (l <= lo <= up <= u):
                return
        y = self.obj[j] / aij
        bound = lr if y >= 0.0 else ur                        # the row bound its multiplier prices
        if not math.isfinite(bound) and y != 0.0 and not self.opt.ReferenceUnboundedSingleton:
            # deviation: the multiplier has the sign of a bound the row does not have -- the dual constraint of column j
            # cannot hold, the LP is dual infeasible (unbounded along x_j if it is feasible at all).
            # free_column_singleton.jl:79 adds y * (+-Inf) to the objective constant here and carries on; the
            # interior-point method then runs into its iteration limit on an objective of -Inf.
            step = math.copysign(1.0, aij) * (1.0 if y < 0.0 else -1.0)
            def pray(sol):
                sol.x[self.new_var_idx[j]] = step
            return self._decided(TRM_DUAL_INFEASIBLE, primal_ray=pray)
        if y != 0.0:
            self.obj0 += y * bound
        else:
            # zero-cost free singleton: nothing is priced (0 * Inf would be NaN when the row is one-sided) and postsolve
            # needs a FINITE row activity to place x_j on: a finite row bound, or 0 for a free row
            lr = ur = lr if math.isfinite(lr) else (ur if math.isfinite(ur) else 0.0)
        rc, rv = [], []
        for j_, a_ in zip(cols, rvals):
            if not self.colflag[j_] or j_ == j:
                continue
            rc.append(int(j_)); rv.append(float(a_))
            self.obj[j_] -= y * a_
            self.nzcol[j_] -= 1
        self.ops.append(FreeColumnSingleton(i, j, lr, ur, aij, y, rc, rv))
        self.rowflag[i] = False
        self.colflag[j] = False
        self.nrow -= 1
        self.ncol -= 1
        self.updated = True

    # -- Presolve.jl:640-707 + dominated_column.jl:8-139 -------------------------------------------
    def _dominated_columns(self, tol=100 * SQRT_EPS):
        # dual bounds from column singletons with one infinite bound
        for j in np.flatnonzero(self.colflag & (self.nzcol == 1)):
            rows, vals = self._col(j)
            nz, i, aij = 0, -1, 0.0
            for i_, a_ in zip(rows, vals):
                if self.rowflag[i_] and a_ != 0.0:
                    nz += 1
                    if nz > 1:
                        break
                    i, aij = int(i_), float(a_)
            if nz != 1 or aij == 0.0:
                continue
            l, u = self.lcol[j], self.ucol[j]
            y_ = self.obj[j] / aij
            if math.isfinite(l) and not math.isfinite(u):     # a_ij y_i <= c_j
                if aij > 0.0: self.uy[i] = min(self.uy[i], y_)
                else: self.ly[i] = max(self.ly[i], y_)
            elif not math.isfinite(l) and math.isfinite(u):   # a_ij y_i >= c_j
                if aij > 0.0: self.ly[i] = max(self.ly[i], y_)
                else: self.uy[i] = min(self.uy[i], y_)
        for j in range(self.n0):
            self._dominated_column(j, tol)
            if self.status != TRM_UNKNOWN:
                break

    def _dominated_column(self, j, tol):
        if not self.colflag[j]:
            return
        rows, vals = self._col(j)
        ls = us = 0.0
        with np.errstate(invalid="ignore"):
            for i, aij in zip(rows, vals):
                if not self.rowflag[i] or aij == 0.0:
                    continue
                ls += aij * (self.ly[i] if aij >= 0.0 else self.uy[i])
                us += aij * (self.uy[i] if aij >= 0.0 else self.ly[i])
        cj = self.obj[j]
        if cj - us > tol:
            bound, ray = self.lcol[j], -1.0                   # reduced cost always positive: lower bound
        elif cj - ls < -tol:
            bound, ray = self.ucol[j], 1.0                    # always negative: upper bound
        else:
            return
        if not math.isfinite(bound):
            def pray(sol):
                sol.x[self.new_var_idx[j]] = ray
            return self._decided(TRM_DUAL_INFEASIBLE, primal_ray=pray)
        self.obj0 += cj * bound
        keep_r, keep_v = [], []
        for i, aij in zip(rows, vals):
            if not self.rowflag[i]:
                continue
            keep_r.append(int(i)); keep_v.append(float(aij))
            self.lrow[i] -= aij * bound
            self.urow[i] -= aij * bound
            self.nzrow[i] -= 1
            if self.nzrow[i] == 1:
                self.row_singletons.append(int(i))
        self.ops.append(DominatedColumn(j, float(bound), float(cj), np.array(keep_r, np.int64), np.array(keep_v)))
        self.colflag[j] = False
        self.ncol -= 1
        self.updated = True

    # -- Presolve.jl:374-452 ---------------------------------------------------------------------
    def run(self):
        def stop():
            return self.status != TRM_UNKNOWN
        self._check_bounds()
        if self.status == TRM_PRIMAL_INFEASIBLE:
            return self.status
        for i in range(self.m0):
            self._drop_empty_row(i)
        for j in range(self.n0):
            self._drop_empty_column(j)
            if stop():
                break
        if stop():
            return self.status
        self.row_singletons = [int(i) for i in np.flatnonzero(self.rowflag & (self.nzrow == 1))]
        self.updated = True
        self.npasses = 0
        while self.updated and not stop():
            self.npasses += 1
            self.updated = False
            steps = (
                self._check_bounds,
                lambda: [self._drop_empty_column(j) for j in range(self.n0) if not stop()],
                self._drop_row_singletons,
                lambda: [self._drop_fixed_variable(j) for j in np.flatnonzero(self.colflag)],
                self._drop_row_singletons,
                lambda: [self._forcing_row(i) for i in np.flatnonzero(self.rowflag)],
                self._drop_row_singletons,
                lambda: [self._free_column_singleton(j) for j in range(self.n0)],
                self._drop_row_singletons,
                self._dominated_columns,
            )
            for step in steps:
                step()
                if stop():
                    return self.status
        for j in range(self.n0):
            self._drop_empty_column(j)
            if stop():
                break
        if self.nrow == 0 and self.ncol == 0 and not stop():  # nothing left: optimal, objective = the constant
            self.status = TRM_OPTIMAL
            sol = self.solution
            sol.resize(0, 0)
            sol.primal_status = sol.dual_status = SLN_OPTIMAL
            sol.is_primal_ray = sol.is_dual_ray = False
            sol.z_primal = sol.z_dual = self.obj0
        self._index_mapping()
        return self.status

    # -- Presolve.jl:177-305 ---------------------------------------------------------------------
    def reduced_problem(self):
        """The LP the interior-point method solves: surviving rows / columns, explicit zeros dropped, maximisation
        restored, then A <- R^-1 A C^-1 with R = sqrt(||row||_2), C = sqrt(||col||_2) of the UNSCALED reduced matrix
        (1 for an empty one); row bounds / R, objective / C, column bounds * C."""
        if self.old_con_idx is None:
            self._index_mapping()
        A = self.A_csc[self.old_con_idx][:, self.old_var_idx].tocsc()
        A.eliminate_zeros()
        A.sort_indices()
        sgn = 1.0 if self.objsense_min else -1.0
        obj = sgn * self.obj[self.old_var_idx]
        obj0 = sgn * self.obj0
        lcon, ucon = self.lrow[self.old_con_idx].copy(), self.urow[self.old_con_idx].copy()
        lvar, uvar = self.lcol[self.old_var_idx].copy(), self.ucol[self.old_var_idx].copy()
        sq = A.multiply(A)
        r = np.sqrt(np.asarray(sq.sum(axis=1)).ravel()); r[r == 0.0] = 1.0
        c = np.sqrt(np.asarray(sq.sum(axis=0)).ravel()); c[c == 0.0] = 1.0
        r, c = np.sqrt(r), np.sqrt(c)
        A = sp.csc_matrix(sp.diags(1.0 / r) @ A @ sp.diags(1.0 / c))
        A.sort_indices()
        self.row_scaling, self.col_scaling = r, c
        self.reduced = LP(A, obj / c, obj0, lcon / r, ucon / r, lvar * c, uvar * c, self.objsense_min, self.lp.name)
        return self.reduced

    # -- Presolve.jl:320-365 ---------------------------------------------------------------------
    def postsolve(self, inner):
        """Solution of the original problem from a solution of the reduced one."""
        if (inner.m, inner.n) != (self.nrow, self.ncol):
            raise ValueError(f"Inner solution has size {(inner.m, inner.n)} but presolved problem has size {(self.nrow, self.ncol)}")
        sol = Solution(self.m0, self.n0)
        sol.primal_status, sol.dual_status = inner.primal_status, inner.dual_status
        sol.is_primal_ray, sol.is_dual_ray = inner.is_primal_ray, inner.is_dual_ray
        sol.z_primal, sol.z_dual = inner.z_primal, inner.z_dual
        cs, rs = self.col_scaling[: self.ncol], self.row_scaling[: self.nrow]
        if self.reduced is None:                              # decided by presolve: never scaled
            cs, rs = np.ones(self.ncol), np.ones(self.nrow)
        sol.x[self.old_var_idx] = inner.x / cs
        sol.s_lower[self.old_var_idx] = inner.s_lower * cs
        sol.s_upper[self.old_var_idx] = inner.s_upper * cs
        sol.y_lower[self.old_con_idx] = inner.y_lower / rs
        sol.y_upper[self.old_con_idx] = inner.y_upper / rs
        for op in reversed(self.ops):
            _UNDO[type(op)](sol, op)
        sol.Ax = self.A_csr @ sol.x
        return sol


def _y(sol, i):
    return sol.y_lower[i] - sol.y_upper[i]


def _undo_empty_row(sol, op):                                # empty_row.jl:70-74
    sol.y_lower[op.i], sol.y_upper[op.i] = _pos(op.y), _neg(op.y)


def _undo_empty_column(sol, op):                             # empty_column.jl:99-104
    sol.x[op.j] = op.x
    sol.s_lower[op.j], sol.s_upper[op.j] = _pos(op.s), _neg(op.s)


def _undo_fixed(sol, op):                                    # fixed_variable.jl:57-66, dominated_column.jl:141-154
    sol.x[op.j] = op.x
    s = 0.0 if sol.is_dual_ray else op.c
    for i, aij in zip(op.rows, op.vals):
        s -= aij * _y(sol, i)
    sol.s_lower[op.j], sol.s_upper[op.j] = _pos(s), _neg(s)


def _undo_row_singleton(sol, op):                            # row_singleton.jl:76-97
    if op.force_lower:
        if op.aij > 0.0: sol.y_lower[op.i] = sol.s_lower[op.j] / op.aij
        else: sol.y_upper[op.i] = sol.s_lower[op.j] / abs(op.aij)
        sol.s_lower[op.j] = 0.0
    if op.force_upper:
        if op.aij > 0.0: sol.y_upper[op.i] = sol.s_upper[op.j] / op.aij
        else: sol.y_lower[op.i] = sol.s_upper[op.j] / abs(op.aij)
        sol.s_upper[op.j] = 0.0


def _undo_dominated_row(sol, op):                            # forcing_row.jl:174-178
    sol.y_lower[op.i] = sol.y_upper[op.i] = 0.0


def _undo_forcing_row(sol, op):                              # forcing_row.jl:181-212
    for j, xj in zip(op.cols, op.xs):
        sol.x[j] = xj
    z = []
    for cj, rws, cvs in zip(op.cs, op.col_rows, op.col_vals):
        zj = cj
        for k, akj in zip(rws, cvs):
            zj -= akj * _y(sol, k)
        z.append(zj)
    ratios = [zj / aij for zj, aij in zip(z, op.vals)]
    # The row multiplier must leave every reduced cost with the sign of the bound its variable sits on:
    # minimal activity == u (recorded at_lower = True): s_j = z_j - a_ij y >= 0 for a_ij > 0 and <= 0 for a_ij < 0,
    # i.e. y <= z_j / a_ij for all j -> the MINIMUM; maximal activity == l: the maximum.
    # deviation: forcing_row.jl:196 takes the maximum when the record says `true` -- with forcing_row.jl:102 storing
    # `true` for the minimal-activity case that returns reduced costs of the wrong sign whenever the row has two or
    # more live entries with different ratios (the function is marked TODO there).  PresolveOptions.
    # ReferenceForcingRowDual = True reproduces it.
    y = (min(ratios) if op.at_lower else max(ratios)) if not op.literal else (max(ratios) if op.at_lower else min(ratios))
    sol.y_lower[op.i], sol.y_upper[op.i] = _pos(y), _neg(y)
    for j, aij, zj in zip(op.cols, op.vals, z):
        s = zj - aij * y
        sol.s_lower[j], sol.s_upper[j] = _pos(s), _neg(s)


def _undo_free_column_singleton(sol, op):                    # free_column_singleton.jl:111-126
    sol.y_lower[op.i], sol.y_upper[op.i] = _pos(op.y), _neg(op.y)
    sol.s_lower[op.j] = sol.s_upper[op.j] = 0.0
    x = 0.0 if sol.is_primal_ray else (op.l if op.y >= 0.0 else op.u)
    for k, aik in zip(op.cols, op.vals):
        x -= aik * sol.x[k]
    sol.x[op.j] = x / op.aij


_UNDO = {EmptyRow: _undo_empty_row, EmptyColumn: _undo_empty_column, FixedVariable: _undo_fixed,
         RowSingleton: _undo_row_singleton, DominatedRow: _undo_dominated_row, ForcingRow: _undo_forcing_row,
         FreeColumnSingleton: _undo_free_column_singleton, DominatedColumn: _undo_fixed}
