"""IPM harness: the CALLER of the KKT backend, restated so that "iterations, objective, residuals"
can be produced without Julia.

Test infrastructure only.  It restates, as a driver of the three-function KKT interface:
  * free-format MPS reading (what /root/reference/src/Interfaces/tulip_julia_api.jl:18-39 gets
    from QPSReader.readqps(mpsformat=:free)),
  * the standard-form conversion of /root/reference/src/IPM/ipmdata.jl:64-173,
  * the homogeneous self-dual loop of /root/reference/src/IPM/HSD/HSD.jl:203-350 and
    /root/reference/src/IPM/HSD/step.jl:10-401 (Mehrotra predictor-corrector + Gondzio's multiple
    centrality corrections, regularisation schedule and the PosDefException retry loop),
  * Mehrotra's predictor-corrector loop of /root/reference/src/IPM/MPC/MPC.jl:101-410 and
    /root/reference/src/IPM/MPC/step.jl:10-358 (starting point from two half-zero solves, clamped
    regularisations, separate primal/dual step lengths, extra centrality corrections),
with Tulip's defaults (/root/reference/src/IPM/options.jl:1-25).  The reader, the standard form and presolve + scaling
are product code since round 2 (tulip.jl_amd/problem.py, presolve.py, model.py); `solve_lp` below runs without
presolve (Presolve_Level = 0 semantics), tests/test_presolve.py runs the loops under the presolving front end.

A backend is anything with update(theta_inv, regP, regD) / solve(dx, dy, xi_p, xi_d) that raises
PosDef on a failed factorisation: the HIP library (HipBackend) or the CPU oracle (OracleBackend).
"""
import math
import time

import numpy as np
import scipy.sparse as sp

SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))
INF = float("inf")


class PosDef(ArithmeticError):
    pass


from tulip_jl_amd.problem import LP, IPMData, read_free_mps, standard_form  # noqa: E402,F401  (product code since round 2)


# ------------------------------------------------------------------------------------------------
# backends
# ------------------------------------------------------------------------------------------------
class HipBackend:
    """KKT.setup/update!/solve! through libtlpk.so (tulip.jl_amd/kkt.py)."""

    def __init__(self, A, **backend_kw):
        import tulip_jl_amd as tk
        self.tk = tk
        self.kkt = tk.setup(A, tk.K1(), tk.Backend(**backend_kw))
        self.name = f"{tk.backend(self.kkt)} / {tk.linear_system(self.kkt)}"

    def update(self, th, rp, rd):
        try:
            self.tk.update(self.kkt, th, rp, rd)
        except self.tk.PosDefException as e:
            raise PosDef(str(e))

    def solve(self, dx, dy, xp, xd):
        self.tk.solve(dx, dy, self.kkt, xp, xd)


class OracleBackend:
    """The CPU oracle behind the same three calls (plumbing config C1, CPU-side comparator)."""

    def __init__(self, A, perm=None):
        from oracle_binding import OracleK1
        self.o = OracleK1(A, perm)
        self.name = "CPU oracle (left-looking Cholesky) / Normal equations (K1)"

    def update(self, th, rp, rd):
        from oracle_binding import OraclePosDefError
        try:
            self.o.update(th, rp, rd)
        except OraclePosDefError as e:
            raise PosDef(str(e))

    def solve(self, dx, dy, xp, xd):
        ddx, ddy = self.o.solve(xp, xd)
        dx[:] = ddx; dy[:] = ddy


class SupernodalBackend:
    """The CHOLMOD-class CPU comparator (oracle/k1_supernodal.c: supernodal multifrontal Cholesky on dpotrf / dtrsm /
    dsyrk, OpenMP over the tree) behind the same three calls: the backend swap at BENCHMARK scale that the 1-core
    simplicial oracle cannot do (SURVEY.md 8(d) parity protocol (ii); callers HSD/step.jl:28-51, MPC/step.jl:28-51).
    Same ordering and supernodes as the HIP run (an analyse-only libtlpk handle supplies the symbolic structure)."""

    def __init__(self, A, threads=0, **backend_kw):
        import tulip_jl_amd as tk
        from oracle_binding import SupernodalK1
        backend_kw = dict(backend_kw); backend_kw["device"] = -1
        self._sym = tk.setup(A, tk.K1(), tk.Backend(**backend_kw))
        self.o = SupernodalK1(A, self._sym, threads=threads)
        self.name = f"CPU supernodal (OpenBLAS, {self.o.threads} threads) / Normal equations (K1)"

    def update(self, th, rp, rd):
        from oracle_binding import OraclePosDefError
        try:
            self.o.update(th, rp, rd)
        except OraclePosDefError as e:
            raise PosDef(str(e))

    def solve(self, dx, dy, xp, xd):
        ddx, ddy = self.o.solve(xp, xd)
        dx[:] = ddx; dy[:] = ddy


# ------------------------------------------------------------------------------------------------
# HSD
# ------------------------------------------------------------------------------------------------
class Options:
    IterationsLimit = 100
    TimeLimit = INF
    TolerancePFeas = ToleranceDFeas = ToleranceRGap = ToleranceIFeas = SQRT_EPS
    CorrectionLimit = 3
    StepDampFactor = 0.9995
    GammaMin = 0.1
    CentralityOutlierThreshold = 0.1
    PRegMin = DRegMin = SQRT_EPS
    OutputLevel = 0


class Point:
    def __init__(self, m, n, p):
        self.m, self.n, self.p = m, n, p
        self.x = np.zeros(n); self.xl = np.zeros(n); self.xu = np.zeros(n)
        self.y = np.zeros(m); self.zl = np.zeros(n); self.zu = np.zeros(n)
        self.tau = 1.0; self.kappa = 1.0; self.mu = 1.0

    def update_mu(self):                       # point.jl:45-48 (hflag = true)
        self.mu = (self.xl @ self.zl + self.xu @ self.zu + self.tau * self.kappa) / (self.p + 1)


def _max_step_vec(x, dx):                      # step.jl:274-287
    neg = dx < 0
    return float(np.min(-x[neg] / dx[neg])) if neg.any() else INF


def _max_step(pt, d):                          # step.jl:294-306
    at = (-pt.tau / d.tau) if d.tau < 0 else 1.0
    ak = (-pt.kappa / d.kappa) if d.kappa < 0 else 1.0
    return min(1.0, _max_step_vec(pt.xl, d.xl), _max_step_vec(pt.xu, d.xu),
               _max_step_vec(pt.zl, d.zl), _max_step_vec(pt.zu, d.zu), at, ak)


class HSD:
    def __init__(self, dat, backend, options=None):
        self.dat, self.kkt, self.opt = dat, backend, options or Options()
        m, n = dat.nrow, dat.ncol
        self.p = int(dat.lflag.sum() + dat.uflag.sum())
        self.pt = Point(m, n, self.p)
        self.regP = np.ones(n); self.regD = np.ones(m); self.regG = 1.0     # HSD.jl:50-52
        self.niter = 0
        self.status = "Trm_Unknown"
        self.primal_status = self.dual_status = "Sln_Unknown"
        self.timers = {"Factorization": 0.0, "KKT": 0.0, "n_update": 0, "n_solve": 0, "n_bump": 0}
        self.log = []

    # HSD.jl:77-128
    def compute_residuals(self):
        pt, d = self.pt, self.dat
        self.rp = pt.tau * d.b - d.A @ pt.x
        self.rl = (-pt.x + pt.xl + pt.tau * d.lz) * d.lflag
        self.ru = (-pt.x - pt.xu + pt.tau * d.uz) * d.uflag
        self.rd = pt.tau * d.c - d.A.T @ pt.y + pt.zu * d.uflag - pt.zl * d.lflag
        dualsum = d.b @ pt.y + d.lz @ pt.zl - d.uz @ pt.zu
        self.rg = pt.kappa + (d.c @ pt.x - dualsum)
        nrm = lambda v: float(np.abs(v).max(initial=0.0))    # noqa: E731
        self.rp_nrm, self.rl_nrm, self.ru_nrm, self.rd_nrm = nrm(self.rp), nrm(self.rl), nrm(self.ru), nrm(self.rd)
        self.rg_nrm = abs(self.rg)
        self.primal_objective = d.c @ pt.x / pt.tau + d.c0
        self.dual_objective = dualsum / pt.tau + d.c0

    # HSD.jl:136-196
    def update_solver_status(self):
        o, pt, d = self.opt, self.pt, self.dat
        nrm = lambda v: float(np.abs(v).max(initial=0.0))    # noqa: E731
        self.status = "Trm_Unknown"
        rho_p = max(self.rp_nrm / (pt.tau * (1 + nrm(d.b))), self.rl_nrm / (pt.tau * (1 + nrm(d.lz))),
                    self.ru_nrm / (pt.tau * (1 + nrm(d.uz))))
        rho_d = self.rd_nrm / (pt.tau * (1 + nrm(d.c)))
        rho_g = abs(self.primal_objective - self.dual_objective) / (1 + abs(self.dual_objective))
        self.rho = (rho_p, rho_d, rho_g)
        self.primal_status = "Sln_FeasiblePoint" if rho_p <= o.TolerancePFeas else "Sln_Unknown"
        self.dual_status = "Sln_FeasiblePoint" if rho_d <= o.ToleranceDFeas else "Sln_Unknown"
        if rho_p <= o.TolerancePFeas and rho_d <= o.ToleranceDFeas and rho_g <= o.ToleranceRGap:
            self.primal_status = self.dual_status = "Sln_Optimal"
            self.status = "Trm_Optimal"
            return
        if max(nrm(d.A @ pt.x), nrm((pt.x - pt.xl) * d.lflag), nrm((pt.x + pt.xu) * d.uflag)) * \
                (nrm(d.c) / max(1.0, nrm(d.b))) < -o.ToleranceIFeas * (d.c @ pt.x):
            self.primal_status = "Sln_InfeasibilityCertificate"
            self.status = "Trm_DualInfeasible"
            return
        delta = d.A.T @ pt.y + pt.zl * d.lflag - pt.zu * d.uflag
        if nrm(delta) * max(nrm(d.lz), nrm(d.uz), nrm(d.b)) / max(1.0, nrm(d.c)) < \
                (d.b @ pt.y + d.lz @ pt.zl - d.uz @ pt.zu) * o.ToleranceIFeas:
            self.dual_status = "Sln_InfeasibilityCertificate"
            self.status = "Trm_PrimalInfeasible"

    def _kkt_solve(self, dx, dy, xp, xd):
        t0 = time.perf_counter()
        self.kkt.solve(dx, dy, np.ascontiguousarray(xp), np.ascontiguousarray(xd))
        self.timers["KKT"] += time.perf_counter() - t0
        self.timers["n_solve"] += 1

    # step.jl:198-266
    def solve_newton_system(self, D, hx, hy, h0, xi_p, xi_l, xi_u, xi_d, xi_g, xi_xzl, xi_xzu, xi_tk):
        pt, d = self.pt, self.dat
        with np.errstate(divide="ignore", invalid="ignore"):
            tl = np.where(d.lflag, (xi_xzl + pt.zl * xi_l) / pt.xl, 0.0)
            tu = np.where(d.uflag, (xi_xzu - pt.zu * xi_u) / pt.xu, 0.0)
            zxl = np.where(d.lflag, pt.zl / pt.xl, 0.0); zxu = np.where(d.uflag, pt.zu / pt.xu, 0.0)
            ixl = np.where(d.lflag, xi_xzl / pt.xl, 0.0); ixu = np.where(d.uflag, xi_xzu / pt.xu, 0.0)
        xi_d_ = xi_d - tl + tu
        self._kkt_solve(D.x, D.y, xi_p, xi_d_)
        xi_g_ = (xi_g + xi_tk / pt.tau - ixl @ d.lz + ixu @ d.uz - (zxl * xi_l) @ d.lz - (zxu * xi_u) @ d.uz)
        D.tau = (xi_g_ + (d.c + zxl * d.lz + zxu * d.uz) @ D.x - d.b @ D.y) / h0
        D.x += D.tau * hx
        D.y += D.tau * hy
        D.xl = (-xi_l + D.x - D.tau * d.lz) * d.lflag
        D.xu = (xi_u - D.x + D.tau * d.uz) * d.uflag
        with np.errstate(divide="ignore", invalid="ignore"):
            D.zl = np.where(d.lflag, (xi_xzl - pt.zl * D.xl) / pt.xl, 0.0)
            D.zu = np.where(d.uflag, (xi_xzu - pt.zu * D.xu) / pt.xu, 0.0)
        D.kappa = (xi_tk - pt.kappa * D.tau) / pt.tau

    # step.jl:325-401
    def compute_higher_corrector(self, Dc, gamma, hx, hy, h0, D, alpha, beta):
        pt, d = self.pt, self.dat
        a_ = min(1.0, 2.0 * alpha)
        vl = ((pt.xl + a_ * D.xl) * (pt.zl + a_ * D.zl)) * d.lflag
        vu = ((pt.xu + a_ * D.xu) * (pt.zu + a_ * D.zu)) * d.uflag
        vt = (pt.tau + a_ * D.tau) * (pt.kappa + a_ * D.kappa)
        mu_l, mu_u = beta * pt.mu * gamma, gamma * pt.mu / beta

        def target(v, flag):
            out = np.where(v < mu_l, mu_l - v, np.where(v > mu_u, mu_u - v, 0.0))
            return np.where(flag, out, v)
        vl = target(vl, d.lflag); vu = target(vu, d.uflag)
        vt = mu_l - vt if vt < mu_l else (mu_u - vt if vt > mu_u else 0.0)
        delta = (vl.sum() + vu.sum() + vt) / (pt.p + 1)
        vl = vl - delta; vu = vu - delta; vt -= delta
        z_m, z_n = np.zeros(pt.m), np.zeros(pt.n)
        self.solve_newton_system(Dc, hx, hy, h0, z_m, z_n, z_n, z_n, 0.0, vl, vu, vt)
        Dc.x += D.x; Dc.xl += D.xl; Dc.xu += D.xu; Dc.y += D.y; Dc.zl += D.zl; Dc.zu += D.zu
        Dc.tau += D.tau; Dc.kappa += D.kappa
        return _max_step(pt, Dc)

    # step.jl:10-151
    def compute_step(self):
        o, pt, d = self.opt, self.pt, self.dat
        with np.errstate(divide="ignore", invalid="ignore"):
            th_l = np.where(d.lflag, pt.zl / pt.xl, 0.0)
            th_u = np.where(d.uflag, pt.zu / pt.xu, 0.0)
        theta_inv = th_l + th_u                       # exactly 0 for free variables
        self.regP = np.maximum(o.PRegMin, self.regP / 10)
        self.regD = np.maximum(o.DRegMin, self.regD / 10)
        self.regG = max(o.PRegMin, self.regG / 10)
        nbump = 0
        while nbump <= 3:
            try:
                t0 = time.perf_counter()
                self.kkt.update(theta_inv, self.regP, self.regD)
                self.timers["Factorization"] += time.perf_counter() - t0
                self.timers["n_update"] += 1
                break
            except PosDef:
                self.regD *= 100; self.regP *= 100; self.regG *= 100
                nbump += 1
                self.timers["n_bump"] += 1
        self.timers["max_bumps_in_a_step"] = max(self.timers.get("max_bumps_in_a_step", 0), nbump)
        if not nbump < 3:                              # step.jl:51 (the reference's off-by-one is kept)
            raise PosDef("factorization could not be saved")
        D, Dc = Point(pt.m, pt.n, pt.p), Point(pt.m, pt.n, pt.p)
        hx, hy = np.zeros(pt.n), np.zeros(pt.m)
        xi_ = d.c - th_l * d.lz - th_u * d.uz
        self._kkt_solve(hx, hy, d.b, xi_)              # xi_p aliases dat.b: inputs must be const
        h0 = (d.lz @ (d.lz * th_l) + d.uz @ (d.uz * th_u) - (d.c + th_l * d.lz + th_u * d.uz) @ hx
              + d.b @ hy + pt.kappa / pt.tau + self.regG)
        self.solve_newton_system(D, hx, hy, h0, self.rp, self.rl, self.ru, self.rd, self.rg,
                                 -(pt.xl * pt.zl) * d.lflag, -(pt.xu * pt.zu) * d.uflag, -pt.tau * pt.kappa)
        alpha = _max_step(pt, D)
        gamma = (1 - alpha) ** 2 * min(1 - alpha, o.GammaMin)
        eta = 1 - gamma
        self.solve_newton_system(D, hx, hy, h0, eta * self.rp, eta * self.rl, eta * self.ru, eta * self.rd,
                                 eta * self.rg,
                                 (-pt.xl * pt.zl + gamma * pt.mu - D.xl * D.zl) * d.lflag,
                                 (-pt.xu * pt.zu + gamma * pt.mu - D.xu * D.zu) * d.uflag,
                                 -pt.tau * pt.kappa + gamma * pt.mu - D.tau * D.kappa)
        alpha = _max_step(pt, D)
        ncor = 0
        while ncor < o.CorrectionLimit and alpha < 0.999:
            a_ = alpha
            ncor += 1
            ac = self.compute_higher_corrector(Dc, gamma, hx, hy, h0, D, a_, o.CentralityOutlierThreshold)
            if ac > a_:
                for k in ("x", "xl", "xu", "y", "zl", "zu"):
                    setattr(D, k, getattr(Dc, k).copy())
                D.tau, D.kappa = Dc.tau, Dc.kappa
                alpha = ac
            if ac < 1.1 * a_:
                break
        alpha *= o.StepDampFactor
        pt.x += alpha * D.x; pt.xl += alpha * D.xl; pt.xu += alpha * D.xu
        pt.y += alpha * D.y; pt.zl += alpha * D.zl; pt.zu += alpha * D.zu
        pt.tau += alpha * D.tau; pt.kappa += alpha * D.kappa
        pt.update_mu()

    # HSD.jl:203-350
    def optimize(self):
        o, pt, d = self.opt, self.pt, self.dat
        tstart = time.perf_counter()
        self.niter = 0
        pt.x[:] = 0; pt.xl[:] = 1.0 * d.lflag; pt.xu[:] = 1.0 * d.uflag
        pt.y[:] = 0; pt.zl[:] = 1.0 * d.lflag; pt.zu[:] = 1.0 * d.uflag
        pt.tau = pt.kappa = 1.0
        pt.update_mu()
        while True:
            self.compute_residuals()
            pt.update_mu()
            eps_ = 1.0 if d.objsense else -1.0
            self.log.append((self.niter, eps_ * self.primal_objective, eps_ * self.dual_objective,
                             max(self.rp_nrm, self.ru_nrm), self.rd_nrm, self.rg_nrm, pt.mu))
            if o.OutputLevel > 0:
                print("%4d  %+14.7e  %+14.7e  %8.2e %8.2e %8.2e  %7.1e" % self.log[-1])
            self.update_solver_status()
            if self.status in ("Trm_Optimal", "Trm_PrimalInfeasible", "Trm_DualInfeasible"):
                break
            if self.niter >= o.IterationsLimit:
                self.status = "Trm_IterationLimit"; break
            if time.perf_counter() - tstart >= o.TimeLimit:
                self.status = "Trm_TimeLimit"; break
            try:
                self.compute_step()
            except PosDef:
                self.status = "Trm_NumericalProblem"; break
            except MemoryError:
                self.status = "Trm_MemoryLimit"; break
            self.niter += 1
        return self

    # model.jl:156-215 (the part the examples assert on)
    def solution(self):
        pt, d = self.pt, self.dat
        ray = "Sln_InfeasibilityCertificate" in (self.primal_status, self.dual_status)
        t_ = 1.0 if ray else 1.0 / pt.tau
        n = d.nvar
        x = pt.x[:n] * t_
        s = (pt.zl[:n] - pt.zu[:n]) * t_
        y = pt.y * t_
        sgn = 1.0 if d.objsense else -1.0
        return {"status": self.status, "niter": self.niter, "x": x, "y": y, "s": s,
                "z_primal": sgn * self.primal_objective, "z_dual": sgn * self.dual_objective,
                "primal_status": self.primal_status, "dual_status": self.dual_status, "rho": self.rho}


# ------------------------------------------------------------------------------------------------
# MPC (Mehrotra predictor-corrector): MPC/MPC.jl:218-410, MPC/step.jl
# ------------------------------------------------------------------------------------------------
def _max_step_pd(pt, d):                       # MPC/step.jl:206-217: separate primal / dual step lengths
    ap = min(1.0, _max_step_vec(pt.xl, d.xl), _max_step_vec(pt.xu, d.xu))
    ad = min(1.0, _max_step_vec(pt.zl, d.zl), _max_step_vec(pt.zu, d.zu))
    return ap, ad


class MPC:
    """Restated caller of the KKT ABI for the non-homogeneous algorithm.  Its call pattern differs
    from HSD's: the starting point costs one update! with theta_inv = 0, regP = 1, regD = 1e-6 and
    two solve! calls with a zero half of the right-hand side (MPC.jl:359-363); every iteration
    is one update! (regularisations /10, clamped to [sqrt(eps), 1], x100 on PosDefException)
    and 2 + ncor solve! calls."""

    def __init__(self, dat, backend, options=None):
        self.dat, self.kkt, self.opt = dat, backend, options or Options()
        m, n = dat.nrow, dat.ncol
        self.p = int(dat.lflag.sum() + dat.uflag.sum())
        self.pt = Point(m, n, self.p)
        self.regP = np.ones(n); self.regD = np.ones(m)          # MPC.jl:73-74
        self.niter = 0
        self.status = "Trm_Unknown"
        self.primal_status = self.dual_status = "Sln_Unknown"
        self.timers = {"Factorization": 0.0, "KKT": 0.0, "n_update": 0, "n_solve": 0, "n_bump": 0}
        self.log = []
        self.alpha_p = self.alpha_d = 0.0

    def _update_mu(self):                       # point.jl:45-48 with hflag = false
        pt = self.pt
        pt.mu = (pt.xl @ pt.zl + pt.xu @ pt.zu) / pt.p

    def _kkt_update(self, th, rp, rd):
        t0 = time.perf_counter()
        self.kkt.update(th, rp, rd)
        self.timers["Factorization"] += time.perf_counter() - t0
        self.timers["n_update"] += 1

    def _kkt_solve(self, dx, dy, xp, xd):
        t0 = time.perf_counter()
        self.kkt.solve(dx, dy, np.ascontiguousarray(xp), np.ascontiguousarray(xd))
        self.timers["KKT"] += time.perf_counter() - t0
        self.timers["n_solve"] += 1

    # MPC.jl:353-410
    def compute_starting_point(self):
        pt, d = self.pt, self.dat
        m, n = pt.m, pt.n
        self._kkt_update(np.zeros(n), np.ones(n), np.full(m, 1e-6))
        self._kkt_solve(np.zeros(n), pt.y, np.zeros(m), d.c)     # y: unused half is a fresh temporary
        self._kkt_solve(pt.x, np.zeros(m), d.b, np.zeros(n))     # x
        dl = np.where(d.lflag, pt.x - d.lz, 0.0)
        du = np.where(d.uflag, d.uz - pt.x, 0.0)
        dxs = 1.0 + max(0.0, -1.5 * dl.min(initial=0.0), -1.5 * du.min(initial=0.0))
        pt.xl = np.where(d.lflag, dl + dxs, 0.0)
        pt.xu = np.where(d.uflag, du + dxs, 0.0)
        z = d.c - d.A.T @ pt.y
        nb = d.lflag.astype(float) + d.uflag.astype(float)
        with np.errstate(divide="ignore", invalid="ignore"):
            pt.zl = np.where(d.lflag, z / nb, 0.0)
            pt.zu = np.where(d.uflag, -z / nb, 0.0)
        dzs = 1.0 + max(0.0, -1.5 * pt.zl.min(initial=0.0), -1.5 * pt.zu.min(initial=0.0))
        pt.zl[d.lflag] += dzs
        pt.zu[d.uflag] += dzs
        pt.tau, pt.kappa = 1.0, 0.0
        mu = pt.xl @ pt.zl + pt.xu @ pt.zu
        ddx = mu / (2 * (pt.zl.sum() + pt.zu.sum()))
        ddz = mu / (2 * (pt.xl.sum() + pt.xu.sum()))
        pt.xl[d.lflag] += ddx; pt.xu[d.uflag] += ddx
        pt.zl[d.lflag] += ddz; pt.zu[d.uflag] += ddz
        self._update_mu()

    # MPC.jl:101-141
    def compute_residuals(self):
        pt, d = self.pt, self.dat
        self.rp = d.b - d.A @ pt.x
        self.rl = ((d.lz + pt.xl) - pt.x) * d.lflag
        self.ru = (d.uz - (pt.x + pt.xu)) * d.uflag
        self.rd = d.c - d.A.T @ pt.y + pt.zu * d.uflag - pt.zl * d.lflag
        nrm = lambda v: float(np.abs(v).max(initial=0.0))    # noqa: E731
        self.rp_nrm, self.rl_nrm, self.ru_nrm, self.rd_nrm = nrm(self.rp), nrm(self.rl), nrm(self.ru), nrm(self.rd)
        self.primal_objective = d.c @ pt.x + d.c0
        self.dual_objective = d.b @ pt.y + d.lz @ pt.zl - d.uz @ pt.zu + d.c0

    # MPC.jl:149-214
    def update_solver_status(self):
        o, pt, d = self.opt, self.pt, self.dat
        nrm = lambda v: float(np.abs(v).max(initial=0.0))    # noqa: E731
        self.status = "Trm_Unknown"
        rho_p = max(self.rp_nrm / (1 + nrm(d.b)), self.rl_nrm / (1 + nrm(d.lz)), self.ru_nrm / (1 + nrm(d.uz)))
        rho_d = self.rd_nrm / (1 + nrm(d.c))
        rho_g = abs(self.primal_objective - self.dual_objective) / (1 + abs(self.primal_objective))
        self.rho = (rho_p, rho_d, rho_g)
        self.primal_status = "Sln_FeasiblePoint" if rho_p <= o.TolerancePFeas else "Sln_Unknown"
        self.dual_status = "Sln_FeasiblePoint" if rho_d <= o.ToleranceDFeas else "Sln_Unknown"
        if rho_p <= o.TolerancePFeas and rho_d <= o.ToleranceDFeas and rho_g <= o.ToleranceRGap:
            self.primal_status = self.dual_status = "Sln_Optimal"
            self.status = "Trm_Optimal"
            return
        if max(nrm(d.A @ pt.x), nrm((pt.x - pt.xl) * d.lflag), nrm((pt.x + pt.xu) * d.uflag)) * \
                (nrm(d.c) / max(1.0, nrm(d.b))) < -o.ToleranceIFeas * (d.c @ pt.x):
            self.primal_status = "Sln_InfeasibilityCertificate"
            self.status = "Trm_DualInfeasible"
            return
        delta = d.A.T @ pt.y + pt.zl * d.lflag - pt.zu * d.uflag
        if nrm(delta) * max(nrm(d.lz), nrm(d.uz), nrm(d.b)) / max(1.0, nrm(d.c)) < \
                (d.b @ pt.y + d.lz @ pt.zl - d.uz @ pt.zu) * o.ToleranceIFeas:
            self.dual_status = "Sln_InfeasibilityCertificate"
            self.status = "Trm_PrimalInfeasible"

    # MPC/step.jl:165-203
    def solve_newton_system(self, D, xi_p, xi_l, xi_u, xi_d, xi_xzl, xi_xzu):
        pt, d = self.pt, self.dat
        with np.errstate(divide="ignore", invalid="ignore"):
            tl = np.where(d.lflag, (xi_xzl + pt.zl * xi_l) / pt.xl, 0.0)
            tu = np.where(d.uflag, (xi_xzu - pt.zu * xi_u) / pt.xu, 0.0)
        self._kkt_solve(D.x, D.y, xi_p, xi_d - tl + tu)
        D.xl = (-xi_l + D.x) * d.lflag
        D.xu = (xi_u - D.x) * d.uflag
        with np.errstate(divide="ignore", invalid="ignore"):
            D.zl = np.where(d.lflag, (xi_xzl - pt.zl * D.xl) / pt.xl, 0.0)
            D.zu = np.where(d.uflag, (xi_xzu - pt.zu * D.xu) / pt.xu, 0.0)
        D.tau = D.kappa = 0.0

    # MPC/step.jl:10-123
    def compute_step(self):
        o, pt, d = self.opt, self.pt, self.dat
        with np.errstate(divide="ignore", invalid="ignore"):
            th_l = np.where(d.lflag, pt.zl / pt.xl, 0.0)
            th_u = np.where(d.uflag, pt.zu / pt.xu, 0.0)
        theta_inv = th_l + th_u
        self.regP = np.clip(self.regP / 10, SQRT_EPS, 1.0)       # step.jl:29-32
        self.regD = np.clip(self.regD / 10, SQRT_EPS, 1.0)
        nbump = 0
        while nbump <= 3:
            try:
                self._kkt_update(theta_inv, self.regP, self.regD)
                break
            except PosDef:
                self.regD *= 100; self.regP *= 100
                nbump += 1
                self.timers["n_bump"] += 1
        self.timers["max_bumps_in_a_step"] = max(self.timers.get("max_bumps_in_a_step", 0), nbump)
        if not nbump < 3:                              # step.jl:51
            raise PosDef("factorization could not be saved")
        D, Dc = Point(pt.m, pt.n, pt.p), Point(pt.m, pt.n, pt.p)
        lf, uf = d.lflag, d.uflag
        # predictor (affine scaling), step.jl:225-241
        xi_p, xi_l, xi_u, xi_d = self.rp.copy(), self.rl.copy(), self.ru.copy(), self.rd.copy()
        self.solve_newton_system(D, xi_p, xi_l, xi_u, xi_d, -(pt.xl * pt.zl) * lf, -(pt.xu * pt.zu) * uf)
        ap, ad = _max_step_pd(pt, D)
        # corrector, step.jl:246-274
        mu_a = (((pt.xl + ap * D.xl) * lf) @ (pt.zl + ad * D.zl) + ((pt.xu + ap * D.xu) * uf) @ (pt.zu + ad * D.zu)) / pt.p
        sigma = min(max((mu_a / pt.mu) ** 3, SQRT_EPS), 1.0 - SQRT_EPS)
        self.solve_newton_system(Dc, xi_p, xi_l, xi_u, xi_d,
                                 (sigma * pt.mu - D.xl * D.zl - pt.xl * pt.zl) * lf,
                                 (sigma * pt.mu - D.xu * D.zu - pt.xu * pt.zu) * uf)
        ap, ad = _max_step_pd(pt, Dc)
        D, Dc = Dc, Point(pt.m, pt.n, pt.p)
        # extra centrality corrections, step.jl:71-109, 279-322
        z_m, z_n = np.zeros(pt.m), np.zeros(pt.n)
        ncor = 0
        while ncor < o.CorrectionLimit:
            ap_, ad_ = min(ap + 0.3, 1.0), min(ad + 0.3, 1.0)
            g = pt.xl @ pt.zl + pt.xu @ pt.zu
            ga = ((pt.xl + ap * D.xl) * lf) @ (pt.zl + ad * D.zl) + ((pt.xu + ap * D.xu) * uf) @ (pt.zu + ad * D.zu)
            mu = (ga / g) * (ga / g) * (ga / pt.p)

            def target(x, dx, z, dz):                  # compute_target!, step.jl:329-358 (gamma = 0.1)
                v = (x + ap_ * dx) * (z + ad_ * dz)
                tmin, tmax = mu * 0.1, mu / 0.1
                return np.where(v < tmin, tmin - v, np.where(v > tmax, tmax - v, 0.0))
            self.solve_newton_system(Dc, z_m, z_n, z_n, z_n, target(pt.xl, D.xl, pt.zl, D.zl), target(pt.xu, D.xu, pt.zu, D.zu))
            for k in ("x", "xl", "xu", "y", "zl", "zu"):
                setattr(Dc, k, getattr(Dc, k) + getattr(D, k))
            apc, adc = _max_step_pd(pt, Dc)
            if apc >= 1.01 * ap and adc >= 1.01 * ad:
                ap, ad = apc, adc
                D, Dc = Dc, Point(pt.m, pt.n, pt.p)
                ncor += 1
            else:
                break
        ap *= o.StepDampFactor; ad *= o.StepDampFactor
        self.alpha_p, self.alpha_d = ap, ad
        pt.x += ap * D.x; pt.xl += ap * D.xl; pt.xu += ap * D.xu
        pt.y += ad * D.y; pt.zl += ad * D.zl; pt.zu += ad * D.zu
        self._update_mu()

    # MPC.jl:218-351
    def optimize(self):
        o, pt, d = self.opt, self.pt, self.dat
        tstart = time.perf_counter()
        self.niter = 0
        self.compute_starting_point()
        while True:
            self.compute_residuals()
            self._update_mu()
            eps_ = 1.0 if d.objsense else -1.0
            self.log.append((self.niter, eps_ * self.primal_objective, eps_ * self.dual_objective,
                             max(self.rp_nrm, self.rl_nrm, self.ru_nrm), self.rd_nrm, float("nan"), pt.mu))
            if o.OutputLevel > 0:
                print("%4d  %+14.7e  %+14.7e  %8.2e %8.2e %8.2e  %7.1e" % self.log[-1])
            self.update_solver_status()
            if self.status in ("Trm_Optimal", "Trm_PrimalInfeasible", "Trm_DualInfeasible"):
                break
            if self.niter >= o.IterationsLimit:
                self.status = "Trm_IterationLimit"; break
            if time.perf_counter() - tstart >= o.TimeLimit:
                self.status = "Trm_TimeLimit"; break
            try:
                self.compute_step()
            except PosDef:
                self.status = "Trm_NumericalProblem"; break
            except MemoryError:
                self.status = "Trm_MemoryLimit"; break
            self.niter += 1
        return self

    def solution(self):                         # tau = 1 throughout: no rescaling
        pt, d = self.pt, self.dat
        n = d.nvar
        sgn = 1.0 if d.objsense else -1.0
        return {"status": self.status, "niter": self.niter, "x": pt.x[:n].copy(), "y": pt.y.copy(),
                "s": (pt.zl[:n] - pt.zu[:n]).copy(),
                "z_primal": sgn * self.primal_objective, "z_dual": sgn * self.dual_objective,
                "primal_status": self.primal_status, "dual_status": self.dual_status, "rho": self.rho}


def solve_lp(lp, backend_factory, options=None, algorithm="hsd"):
    """load -> standard form -> KKT.setup -> HSD (default, model.jl) or MPC -> solution
    (optimize! without presolve)."""
    dat = standard_form(lp)
    be = backend_factory(dat.A)
    ipm = (HSD if algorithm == "hsd" else MPC)(dat, be, options).optimize()
    return ipm, ipm.solution()
