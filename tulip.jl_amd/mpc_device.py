"""Mehrotra predictor-corrector loop with the iterate resident in HBM.

Host side of the optional device-resident optimizer for Tulip's non-homogeneous algorithm: the control flow of
/root/reference/src/IPM/MPC/MPC.jl:218-351 and /root/reference/src/IPM/MPC/step.jl:10-123, every vector operation
one call into libtlpk.so (`tlpk_mpc_*` and the shared `tlpk_ipm_*`, include/tlpk.h); only scalars cross PCIe.
Tulip's defaults (/root/reference/src/IPM/options.jl:1-25).

    opt = DeviceMPC(A, b, c, l, u, c0=0.0).optimize()
    opt.status, opt.niter, opt.primal_objective, opt._get(0, opt.n)
"""
import time

import numpy as np

from . import _lib
from .hsd_device import INF, SQRT_EPS, DeviceHSD, Options
from .kkt import OutOfMemoryError, PosDefException


class DeviceMPC(DeviceHSD):
    """Shares the vectors on the device, the residual / status quantities and the accessors with DeviceHSD."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.tau, self.kappa = 1.0, 0.0
        self.alpha_p = self.alpha_d = 0.0

    # MPC.jl:101-141: the HSD residual kernels with tau = 1
    def compute_residuals(self):
        o = self._out
        self._call(self.L.tlpk_ipm_residuals(self.kkt._h, 1.0, _lib.as_pd(o)))
        (self.rp_nrm, self.rl_nrm, self.ru_nrm, self.rd_nrm, self.cx, by, lzzl, uzzu, self.xz,
         self.ax_nrm, self.xxl_nrm, self.xxu_nrm, self.delta_nrm) = (float(v) for v in o[:13])
        self.dualsum = by + lzzl - uzzu
        self.primal_objective = self.cx + self.c0
        self.dual_objective = self.dualsum + self.c0
        self.mu = self.xz / self.p if self.p else 0.0                         # point.jl:45-48 without the tau-kappa term

    # MPC.jl:149-214
    def update_solver_status(self):
        o = self.opt
        self.status = "Trm_Unknown"
        rho_p = max(self.rp_nrm / (1 + self.nb), self.rl_nrm / (1 + self.nlz), self.ru_nrm / (1 + self.nuz))
        rho_d = self.rd_nrm / (1 + self.nc)
        rho_g = abs(self.primal_objective - self.dual_objective) / (1 + abs(self.primal_objective))
        self.rho = (rho_p, rho_d, rho_g)
        self.primal_status = "Sln_FeasiblePoint" if rho_p <= o.TolerancePFeas else "Sln_Unknown"
        self.dual_status = "Sln_FeasiblePoint" if rho_d <= o.ToleranceDFeas else "Sln_Unknown"
        if rho_p <= o.TolerancePFeas and rho_d <= o.ToleranceDFeas and rho_g <= o.ToleranceRGap:
            self.primal_status = self.dual_status = "Sln_Optimal"
            self.status = "Trm_Optimal"
            return
        if max(self.ax_nrm, self.xxl_nrm, self.xxu_nrm) * (self.nc / max(1.0, self.nb)) < -o.ToleranceIFeas * self.cx:
            self.primal_status = "Sln_InfeasibilityCertificate"
            self.status = "Trm_DualInfeasible"
            return
        if self.delta_nrm * max(self.nlz, self.nuz, self.nb) / max(1.0, self.nc) < self.dualsum * o.ToleranceIFeas:
            self.dual_status = "Sln_InfeasibilityCertificate"
            self.status = "Trm_PrimalInfeasible"

    def compute_starting_point(self):                                         # MPC.jl:353-410
        self._call(self.L.tlpk_mpc_start(self.kkt._h, _lib.as_pd(self._out)))
        self.timers["n_update"] += 1
        self.timers["n_solve"] += 2
        self.mu = float(self._out[0]) / self.p if self.p else 0.0

    def _newton(self, mode, gmu=0.0):                                         # MPC/step.jl:164-217
        self._call(self.L.tlpk_mpc_newton(self.kkt._h, mode, gmu, _lib.as_pd(self._out)))
        self.timers["n_solve"] += 1
        return min(1.0, float(self._out[0])), min(1.0, float(self._out[1]))

    # MPC/step.jl:10-123
    def compute_step(self):
        o = self.opt
        self.regP = min(max(self.regP / 10, SQRT_EPS), 1.0)                   # step.jl:29-32 (uniform vectors)
        self.regD = min(max(self.regD / 10, SQRT_EPS), 1.0)
        nbump = 0
        while nbump <= 3:
            try:
                self._call(self.L.tlpk_ipm_factor(self.kkt._h, self.regP, self.regD))
                self.timers["n_update"] += 1
                break
            except PosDefException:
                self.regD *= 100; self.regP *= 100
                nbump += 1
                self.timers["n_bump"] += 1
        self.timers["max_bumps_in_a_step"] = max(self.timers.get("max_bumps_in_a_step", 0), nbump)
        if not nbump < 3:                                                     # step.jl:51 (the reference's off-by-one is kept)
            raise PosDefException(0)
        out = self._out
        # predictor (affine scaling), step.jl:225-241
        ap, ad = self._newton(0)
        # corrector, step.jl:246-274
        self._call(self.L.tlpk_mpc_gap(self.kkt._h, ap, ad, _lib.as_pd(out)))
        mu_a = float(out[0]) / self.p
        sigma = min(max((mu_a / self.mu) ** 3, SQRT_EPS), 1.0 - SQRT_EPS)
        ap, ad = self._newton(1, sigma * self.mu)
        # extra centrality corrections, step.jl:71-109, 279-322 (delta = 0.3, gamma = 0.1)
        ncor = 0
        while ncor < o.CorrectionLimit:
            ap_, ad_ = min(ap + 0.3, 1.0), min(ad + 0.3, 1.0)
            self._call(self.L.tlpk_mpc_gap(self.kkt._h, ap, ad, _lib.as_pd(out)))
            ga, g = float(out[0]), float(out[1])
            mu = (ga / g) * (ga / g) * (ga / self.p)
            self._call(self.L.tlpk_mpc_targets(self.kkt._h, ap_, ad_, mu * 0.1, mu / 0.1))
            apc, adc = self._newton(2)
            if apc >= 1.01 * ap and adc >= 1.01 * ad:
                self._call(self.L.tlpk_ipm_accept(self.kkt._h))
                ap, ad = apc, adc
                ncor += 1
            else:
                break
        ap *= o.StepDampFactor; ad *= o.StepDampFactor
        self.alpha_p, self.alpha_d = ap, ad
        self._call(self.L.tlpk_mpc_advance(self.kkt._h, ap, ad, _lib.as_pd(out)))
        self.mu = float(out[0]) / self.p if self.p else 0.0

    # MPC.jl:218-351
    def optimize(self):
        o = self.opt
        tstart = time.perf_counter()
        self.niter = 0
        self.regP = self.regD = 1.0                                           # MPC.jl:73-74
        self.compute_starting_point()
        while True:
            self.compute_residuals()
            self.update_solver_status()
            if self.status in ("Trm_Optimal", "Trm_PrimalInfeasible", "Trm_DualInfeasible"):
                break
            if self.niter >= o.IterationsLimit:
                self.status = "Trm_IterationLimit"; break
            if time.perf_counter() - tstart >= o.TimeLimit:
                self.status = "Trm_TimeLimit"; break
            try:
                self.compute_step()
            except PosDefException:
                self.status = "Trm_NumericalProblem"; break
            except OutOfMemoryError:
                self.status = "Trm_MemoryLimit"; break
            self.niter += 1
        self.seconds = time.perf_counter() - tstart
        return self


__all__ = ["DeviceMPC", "Options", "INF"]
