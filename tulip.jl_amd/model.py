"""Stand-alone front door: read a free-MPS file, presolve + scale, run the interior-point method on the MI355X,
postsolve (the flow of /root/reference/src/model.jl:67-166).

    m = Model.load("problem.mps")            # tulip_julia_api.jl:18-39
    m.optimize()                             # model.jl:67-166: presolve -> standard form -> IPM -> postsolve
    m.status, m.objective_value(), m.dual_objective_value(), m.solution.x, m.solution.y_lower ...

The interior-point method is `DeviceHSD` (hsd_device.py, Tulip's default homogeneous self-dual loop) or `DeviceMPC`
(mpc_device.py, `algorithm="mpc"`), both with the iterate in HBM and every Newton step through libtlpk.so.  `optimize(ipm=...)` accepts any callable `LP -> InnerResult`
instead (the parity tests run the same front end over the CPU oracle backend that way); the product default
needs the GPU and fails loudly without it.
"""
from dataclasses import dataclass

import numpy as np

from .presolve import (SLN_FEASIBLE, SLN_OPTIMAL, SLN_RAY, TRM_DUAL_INFEASIBLE, TRM_OPTIMAL, TRM_PRIMAL_INFEASIBLE,
                       Presolve, PresolveOptions, Solution)
from .problem import LP, read_free_mps, standard_form

INF = float("inf")


@dataclass
class InnerResult:
    """What the front end needs from an interior-point run on a standard-form problem (HSD.jl / MPC.jl state)."""
    status: str
    primal_status: str
    dual_status: str
    x: np.ndarray            # standard-form columns (structural first, then slacks)
    y: np.ndarray
    zl: np.ndarray
    zu: np.ndarray
    tau: float
    primal_objective: float
    dual_objective: float
    niter: int = 0


def device_ipm(lp, algorithm="hsd", **backend_kw):
    """Default interior-point run on the standard form of `lp`: DeviceHSD (Tulip's default, model.jl IPM.Factory)
    or DeviceMPC, iterate in HBM."""
    if algorithm == "mpc":
        from .mpc_device import DeviceMPC as Opt
    else:
        from .hsd_device import DeviceHSD as Opt
    d = standard_form(lp)
    opt = Opt(d.A, d.b, d.c, d.l, d.u, c0=d.c0, objsense_min=d.objsense, **backend_kw)
    opt.optimize()
    return InnerResult(opt.status, opt.primal_status, opt.dual_status, opt._get(0, opt.n), opt._get(5, opt.m),
                       opt._get(3, opt.n), opt._get(4, opt.n), opt.tau, opt.primal_objective, opt.dual_objective,
                       opt.niter)


def extract_solution(lp, res):
    """model.jl:168-230: solution of `lp` (rows m, structural columns n) from the interior-point state."""
    m, n = lp.A.shape
    sol = Solution(m, n)
    sol.primal_status, sol.dual_status = res.primal_status, res.dual_status
    sol.is_primal_ray = res.primal_status == SLN_RAY
    sol.is_dual_ray = res.dual_status == SLN_RAY
    t_ = 1.0 if (sol.is_primal_ray or sol.is_dual_ray) else 1.0 / res.tau
    sol.x = np.asarray(res.x[:n], float) * t_
    sol.s_lower = np.asarray(res.zl[:n], float) * t_
    sol.s_upper = np.asarray(res.zu[:n], float) * t_
    y = np.asarray(res.y, float)
    sol.y_lower = np.maximum(y, 0.0) * t_
    sol.y_upper = np.maximum(-y, 0.0) * t_
    sol.Ax = lp.A @ sol.x
    if sol.primal_status == SLN_RAY:
        sol.z_primal = sol.z_dual = -INF
    elif sol.primal_status in (SLN_OPTIMAL, SLN_FEASIBLE):
        sol.z_primal = res.primal_objective
    else:
        sol.z_primal = float("nan")
    if sol.dual_status == SLN_RAY:
        sol.z_primal = sol.z_dual = INF
    elif sol.dual_status in (SLN_OPTIMAL, SLN_FEASIBLE):
        sol.z_dual = res.dual_objective
    else:
        sol.z_dual = float("nan")            # (also after an unbounded ray set it to -inf: model.jl:221-228 as written)
    return sol


class Model:
    def __init__(self, lp=None, presolve_level=1, algorithm="hsd", **backend_kw):
        self.lp = lp
        self.algorithm = algorithm
        self.presolve_options = PresolveOptions(Level=presolve_level)
        self.backend_kw = backend_kw
        self.presolve = None
        self.solution = None
        self.status = "Trm_NotCalled"
        self.inner = None

    @classmethod
    def load(cls, path, **kw):
        return cls(read_free_mps(path), **kw)

    def optimize(self, ipm=None):
        """model.jl:67-166."""
        lp = self.lp
        ipm = ipm or (lambda p: device_ipm(p, self.algorithm, **self.backend_kw))
        lp_inner = lp
        if self.presolve_options.Level > 0:
            ps = self.presolve = Presolve(lp, self.presolve_options)
            st = ps.run()
            self.status = st
            if st in (TRM_OPTIMAL, TRM_PRIMAL_INFEASIBLE, TRM_DUAL_INFEASIBLE):    # presolve solved the problem
                self.solution = ps.postsolve(ps.solution)
                self.inner = None
                return self
            lp_inner = ps.reduced_problem()
        res = self.inner = ipm(lp_inner)
        sol_inner = extract_solution(lp_inner, res)
        self.solution = self.presolve.postsolve(sol_inner) if self.presolve_options.Level > 0 else sol_inner
        self.status = res.status
        return self

    # tulip_julia_api.jl:243-300
    def objective_value(self):
        sol = self.solution
        if sol is None:
            raise RuntimeError("Model has no solution")
        if sol.primal_status == "Sln_Unknown":
            return 0.0
        return float(sol.x @ self.lp.obj) + (0.0 if sol.is_primal_ray else self.lp.obj0)

    def dual_objective_value(self):
        sol, lp = self.solution, self.lp
        if sol is None:
            raise RuntimeError("Model has no solution")
        if sol.dual_status == "Sln_Unknown":
            return 0.0
        fin = lambda v: np.where(np.isfinite(v), v, 0.0)      # noqa: E731
        z = sol.y_lower @ fin(lp.lcon) - sol.y_upper @ fin(lp.ucon) + sol.s_lower @ fin(lp.lvar) - sol.s_upper @ fin(lp.uvar)
        z = z if lp.objsense_min else -z
        return float(z) + (0.0 if sol.is_dual_ray else lp.obj0)
