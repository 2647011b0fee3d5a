"""Homogeneous self-dual interior-point loop with the iterate resident in HBM.

Host side of the optional device-resident optimizer: the control flow of
/root/reference/src/IPM/HSD/HSD.jl:203-350 and /root/reference/src/IPM/HSD/step.jl:10-151, with every
vector operation replaced by one call into libtlpk.so (`tlpk_ipm_*`, include/tlpk.h).  Only scalars
cross the PCIe link: norms and dot products come back, tau, kappa, the regularisations and the step
lengths go in.  Tulip's defaults (/root/reference/src/IPM/options.jl:1-25); no presolve, no scaling.

    opt = DeviceHSD(A, b, c, l, u, c0=0.0)     # standard form: min c'x + c0, A x = b, l <= x <= u
    opt.optimize()
    opt.status, opt.niter, opt.primal_objective, opt.x(), opt.y()
"""
import ctypes as C
import math
import time

import numpy as np

from . import _lib
from .kkt import K1, K2, Backend, DimensionMismatch, OutOfMemoryError, PosDefException, _raise_for, setup

SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))
INF = float("inf")


class Options:
    """/root/reference/src/IPM/options.jl:6-22"""
    IterationsLimit = 100
    TimeLimit = INF
    TolerancePFeas = ToleranceDFeas = ToleranceRGap = ToleranceIFeas = SQRT_EPS
    CorrectionLimit = 3
    StepDampFactor = 0.9995
    GammaMin = 0.1
    CentralityOutlierThreshold = 0.1
    PRegMin = DRegMin = SQRT_EPS


class DeviceHSD:
    def __init__(self, A, b, c, l, u, c0=0.0, objsense_min=True, options=None, system="K1", pair_solves=True, overlap_root=False, **backend_kw):
        # pair_solves: the h-system and the predictor share one pass over the factor (tlpk_ipm_hsolve_newton; same arithmetic)
        # overlap_root: the factorisation does not wait for its status before that pair is enqueued (tlpk_ipm_factor_hsolve_newton);
        # off by default: the root front's serial chain beside the chip-filling sweeps measured 1.1 ms slower per step on C4
        self.pair_solves = bool(pair_solves)
        self.overlap_root = bool(overlap_root)
        # system: "K1" normal equations | "K2" augmented system (the reference's default for Float64, KKT.jl:134-141)
        # one device, or ngpus > 1 (a block-angular LP on one multi-device handle, K1 or K2: every shard keeps the sub-LP of its diagonal
        # blocks on its device, the library reduces the root panel / root right-hand side and the host adds the shards' scalars).
        # Sharded handles (nranks > 1) leave their reductions to the caller, the loops cannot drive them.
        if int(backend_kw.get("nranks", 1)) > 1:
            raise ValueError("the device-resident interior-point loops need one handle for the whole LP: nranks must be 1 "
                             "(ngpus > 1 is fine; sharded handles serve the split-phase KKT.update! / KKT.solve!)")
        self.kkt = setup(A, K2() if str(system).upper() == "K2" else K1(), Backend(**backend_kw))
        self.m, self.n = self.kkt.m, self.kkt.n
        self.opt = options or Options()
        self._b = np.ascontiguousarray(b, dtype=np.float64); self._c = np.ascontiguousarray(c, dtype=np.float64)
        l = np.ascontiguousarray(l, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
        if self._b.shape != (self.m,) or self._c.shape != (self.n,) or l.shape != (self.n,) or u.shape != (self.n,):
            raise DimensionMismatch("b, c, l, u do not match A")
        self.c0, self.objsense = float(c0), bool(objsense_min)
        self.L = _lib.lib()
        self._call(self.L.tlpk_ipm_load(self.kkt._h, _lib.as_pd(self._b), _lib.as_pd(self._c), _lib.as_pd(l), _lib.as_pd(u)))
        lf, uf = np.isfinite(l), np.isfinite(u)
        self.p = int(lf.sum() + uf.sum())                                    # HSD.jl:39
        nrm = lambda v: float(np.abs(v).max(initial=0.0))                    # noqa: E731
        self.nb, self.nc = nrm(self._b), nrm(self._c)
        self.nlz, self.nuz = nrm(np.where(lf, l, 0.0)), nrm(np.where(uf, u, 0.0))
        self.regP = self.regD = self.regG = 1.0                              # HSD.jl:50-52 (uniform vectors)
        self.tau = self.kappa = 1.0
        self.mu = 1.0
        self.niter = 0
        self.status = "Trm_Unknown"
        self.primal_status = self.dual_status = "Sln_Unknown"
        self.timers = {"n_update": 0, "n_solve": 0, "n_bump": 0}
        self._out = np.zeros(16)
        self._sc = np.zeros(8)

    def _call(self, rc):
        if rc == _lib.NOT_POSDEF:
            raise PosDefException(0)
        _raise_for(rc, self.kkt._h, "tlpk_ipm: ")

    # HSD.jl:77-128
    def compute_residuals(self):
        o = self._out
        self._call(self.L.tlpk_ipm_residuals(self.kkt._h, self.tau, _lib.as_pd(o)))
        (self.rp_nrm, self.rl_nrm, self.ru_nrm, self.rd_nrm, self.cx, by, lzzl, uzzu, self.xz,
         self.ax_nrm, self.xxl_nrm, self.xxu_nrm, self.delta_nrm) = (float(v) for v in o[:13])
        self.dualsum = by + lzzl - uzzu
        self.rg = self.kappa + (self.cx - self.dualsum)
        self.rg_nrm = abs(self.rg)
        self.primal_objective = self.cx / self.tau + self.c0
        self.dual_objective = self.dualsum / self.tau + self.c0
        self.mu = (self.xz + self.tau * self.kappa) / (self.p + 1)           # point.jl:45-48

    # HSD.jl:136-196
    def update_solver_status(self):
        o, tau = self.opt, self.tau
        self.status = "Trm_Unknown"
        rho_p = max(self.rp_nrm / (tau * (1 + self.nb)), self.rl_nrm / (tau * (1 + self.nlz)), self.ru_nrm / (tau * (1 + self.nuz)))
        rho_d = self.rd_nrm / (tau * (1 + self.nc))
        rho_g = abs(self.primal_objective - self.dual_objective) / (1 + abs(self.dual_objective))
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

    def _newton(self, mode, xi_g, xi_tk, eta=0.0, gmu=0.0, delta=0.0):
        """step.jl:198-266 on the device; returns (dtau, dkappa, max step over the vector part)."""
        self._sc[:] = (self.tau, self.kappa, self.h0, xi_g, xi_tk, eta, gmu, delta)
        self._call(self.L.tlpk_ipm_newton(self.kkt._h, mode, _lib.as_pd(self._sc), _lib.as_pd(self._out)))
        self.timers["n_solve"] += 1
        return float(self._out[0]), float(self._out[1]), float(self._out[2])

    def _max_step(self, a_vec, dtau, dkappa):                                # step.jl:294-306
        at = (-self.tau / dtau) if dtau < 0 else 1.0
        ak = (-self.kappa / dkappa) if dkappa < 0 else 1.0
        return min(1.0, a_vec, at, ak)

    # step.jl:10-151
    def compute_step(self):
        o = self.opt
        self.regP = max(o.PRegMin, self.regP / 10); self.regD = max(o.DRegMin, self.regD / 10); self.regG = max(o.PRegMin, self.regG / 10)
        nbump = 0
        fused = self.pair_solves and self.overlap_root
        while nbump <= 3:
            try:
                if fused:
                    # factorisation WITHOUT the wait for its status + the h-system / predictor pair: the block-level forward sweeps of
                    # the pair overlap the root front's factorisation; a failed factorisation is reported here, like tlpk_ipm_factor's
                    self._sc[:] = (self.tau, self.kappa, self.regG, self.rg, -self.tau * self.kappa, 0.0, 0.0, 0.0)
                    self._call(self.L.tlpk_ipm_factor_hsolve_newton(self.kkt._h, self.regP, self.regD, _lib.as_pd(self._sc), _lib.as_pd(self._out)))
                else:
                    self._call(self.L.tlpk_ipm_factor(self.kkt._h, self.regP, self.regD))
                self.timers["n_update"] += 1
                break
            except PosDefException:
                self.regD *= 100; self.regP *= 100; self.regG *= 100
                nbump += 1
                self.timers["n_bump"] += 1
        self.timers["max_bumps_in_a_step"] = max(self.timers.get("max_bumps_in_a_step", 0), nbump)
        if not nbump < 3:                                                    # step.jl:51 (the reference's off-by-one is kept)
            raise PosDefException(0)
        if fused:
            self.timers["n_solve"] += 2; self.timers["n_paired"] = self.timers.get("n_paired", 0) + 1
            dtau, dkappa, av, self.h0 = (float(v) for v in self._out[:4])
        elif self.pair_solves:
            # h-system (step.jl:56-76) and predictor: independent right-hand sides, one pass over the factor
            self._sc[:] = (self.tau, self.kappa, self.regG, self.rg, -self.tau * self.kappa, 0.0, 0.0, 0.0)
            self._call(self.L.tlpk_ipm_hsolve_newton(self.kkt._h, _lib.as_pd(self._sc), _lib.as_pd(self._out)))
            self.timers["n_solve"] += 2; self.timers["n_paired"] = self.timers.get("n_paired", 0) + 1
            dtau, dkappa, av, self.h0 = (float(v) for v in self._out[:4])
        else:
            self._call(self.L.tlpk_ipm_hsolve(self.kkt._h, _lib.as_pd(self._out)))
            self.timers["n_solve"] += 1
            self.h0 = float(self._out[0]) + self.kappa / self.tau + self.regG
            # predictor
            dtau, dkappa, av = self._newton(0, self.rg, -self.tau * self.kappa)
        alpha = self._max_step(av, dtau, dkappa)
        gamma = (1 - alpha) ** 2 * min(1 - alpha, o.GammaMin)
        eta = 1 - gamma
        # corrector (second-order terms from the predictor direction, which it overwrites)
        dtau, dkappa, av = self._newton(1, eta * self.rg, -self.tau * self.kappa + gamma * self.mu - dtau * dkappa,
                                        eta=eta, gmu=gamma * self.mu)
        alpha = self._max_step(av, dtau, dkappa)
        ncor = 0
        while ncor < o.CorrectionLimit and alpha < 0.999:                    # step.jl:104-136
            a_ = alpha
            ncor += 1
            # compute_higher_corrector, step.jl:325-401
            beta = o.CentralityOutlierThreshold
            aa = min(1.0, 2.0 * a_)
            mu_l, mu_u = beta * self.mu * gamma, gamma * self.mu / beta
            self._call(self.L.tlpk_ipm_targets(self.kkt._h, aa, mu_l, mu_u, _lib.as_pd(self._out)))
            svl, svu = float(self._out[0]), float(self._out[1])
            vt = (self.tau + aa * dtau) * (self.kappa + aa * dkappa)
            vt = mu_l - vt if vt < mu_l else (mu_u - vt if vt > mu_u else 0.0)
            delta = (svl + svu + vt) / (self.p + 1)
            ctau, ckappa, av = self._newton(2, 0.0, vt - delta, delta=delta)
            ctau += dtau; ckappa += dkappa
            ac = self._max_step(av, ctau, ckappa)
            if ac > a_:
                self._call(self.L.tlpk_ipm_accept(self.kkt._h))
                dtau, dkappa, alpha = ctau, ckappa, ac
            if ac < 1.1 * a_:
                break
        alpha *= o.StepDampFactor
        self._call(self.L.tlpk_ipm_advance(self.kkt._h, alpha, _lib.as_pd(self._out)))
        self.tau += alpha * dtau; self.kappa += alpha * dkappa
        self.mu = (float(self._out[0]) + self.tau * self.kappa) / (self.p + 1)

    # HSD.jl:203-350
    def optimize(self):
        o = self.opt
        tstart = time.perf_counter()
        self._call(self.L.tlpk_ipm_reset(self.kkt._h))
        self.tau = self.kappa = 1.0
        self.regP = self.regD = self.regG = 1.0
        self.niter = 0
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

    def _get(self, what, length):
        v = np.empty(length)
        self._call(self.L.tlpk_ipm_get(self.kkt._h, what, _lib.as_pd(v), length))
        return v

    def solution(self, nvar=None):
        """model.jl:156-215 (the part the examples assert on): x, y, s = zl - zu, rescaled by 1/tau
        unless the point is an infeasibility certificate."""
        ray = "Sln_InfeasibilityCertificate" in (self.primal_status, self.dual_status)
        t_ = 1.0 if ray else 1.0 / self.tau
        n = self.n if nvar is None else nvar
        x = self._get(0, self.n)[:n] * t_
        s = (self._get(3, self.n)[:n] - self._get(4, self.n)[:n]) * t_
        y = self._get(5, self.m) * t_
        sgn = 1.0 if self.objsense else -1.0
        return {"status": self.status, "niter": self.niter, "x": x, "y": y, "s": s,
                "z_primal": sgn * self.primal_objective, "z_dual": sgn * self.dual_objective,
                "primal_status": self.primal_status, "dual_status": self.dual_status, "rho": self.rho}
