"""Shared test helpers: fixtures loading, instance generators, residual checks."""
import glob
import json
import os

import numpy as np
import scipy.sparse as sp

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))


def load_golden(name=None):
    files = sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.json")))
    out = []
    for f in files:
        d = json.load(open(f))
        if name is None or d["name"] == name:
            for k in ("A", "theta_inv", "regP", "regD", "xi_p", "xi_d", "dx", "dy"):
                d[k] = np.array(d[k], dtype=np.float64)
            d["A_csc"] = sp.csc_matrix(d["A"])
            out.append(d)
    return out


def golden_tol(g):
    """10*eps*cond(S), floor 1e-13, relative to the solution size (see test_oracle.py)."""
    return max(1e-13, 10 * np.finfo(float).eps * g["cond_S"])


def kkt_residuals(A, th, rp, rd, xp, xd, dx, dy):
    """The two residual norms of /root/reference/src/KKT/Test/test.jl:39-43."""
    r_p = A @ dx + rd * dy - xp
    r_d = -dx * (th + rp) + A.T @ dy - xd
    return float(np.abs(r_p).max(initial=0.0)), float(np.abs(r_d).max(initial=0.0))


def random_lp_matrix(m, n, nnz_per_col, seed, slack=False):
    """Random sparse m x n matrix (CSC), nnz_per_col distinct rows per column, N(0,1) values;
    optionally [A0 I] (inequality rows -> slack columns, ipmdata.jl:90-96)."""
    rng = np.random.default_rng(seed)
    rows = np.concatenate([rng.choice(m, size=min(nnz_per_col, m), replace=False) for _ in range(n)])
    k = min(nnz_per_col, m)
    cols = np.repeat(np.arange(n), k)
    vals = rng.standard_normal(n * k)
    A = sp.csc_matrix((vals, (rows, cols)), shape=(m, n))
    if slack:
        A = sp.hstack([A, sp.identity(m, format="csc")], format="csc")
    A.sort_indices()
    return A


def ipm_like_data(m, n, seed, regime="mid"):
    rng = np.random.default_rng(seed)
    if regime == "mid":
        th = 10.0 ** rng.uniform(-3, 3, n); rp = np.full(n, 1e-4); rd = np.full(m, 1e-4)
    elif regime == "late":
        th = 10.0 ** rng.uniform(-8, 8, n); th[rng.random(n) < 0.05] = 0.0
        rp = np.full(n, SQRT_EPS); rd = np.full(m, SQRT_EPS)
    else:
        th = np.ones(n); rp = np.ones(n); rd = np.ones(m)
    return th, rp, rd, rng.standard_normal(m), rng.standard_normal(n)


def block_angular(nblocks, mk, nk, m0, nnz_in, link_prob, seed):
    """Block-angular A = [diag(A_1..A_K); B_1 .. B_K]:
    A_k is mk x nk with nnz_in nonzeros per column, each column also hits one linking row
    w.p. link_prob.  Returns (A csc, row_block) with row_block[i] = k for block rows and
    -1 for the m0 linking rows (placed last)."""
    blocks, links = [], []
    for k in range(nblocks):
        rng = np.random.default_rng(seed + k)
        Ak = random_lp_matrix(mk, nk, nnz_in, seed + 1000 + k)
        has = rng.random(nk) < link_prob
        cols = np.nonzero(has)[0]
        rows = rng.integers(0, max(m0, 1), size=cols.size)
        Bk = sp.csc_matrix((rng.standard_normal(cols.size), (rows, cols)), shape=(m0, nk))
        blocks.append(Ak); links.append(Bk)
    top = sp.block_diag(blocks, format="csc")
    A = sp.vstack([top, sp.hstack(links, format="csc")], format="csc") if m0 > 0 else top
    A.sort_indices()
    row_block = np.concatenate([np.repeat(np.arange(nblocks), mk), np.full(m0, -1)]).astype(np.int64)
    return A, row_block


class DevBuf:
    """A device buffer of doubles for the device-pointer entry points, through the HIP runtime libtlpk.so itself is linked
    against (ctypes on libamdhip64.so: the already-loaded instance).  torch is NOT used inside the pytest process: its wheel
    bundles its own HIP runtime, and the second runtime of a process sees no GPU ("No HIP GPUs are available") once libtlpk has
    initialised the first."""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import ctypes
            import tulip_jl_amd._lib as L
            L.lib()                                             # libtlpk.so first: pulls in its libamdhip64.so
            h = ctypes.CDLL("libamdhip64.so")
            h.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
            h.hipFree.argtypes = [ctypes.c_void_p]
            h.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            h.hipDeviceSynchronize.argtypes = []
            cls._hip = h
        return cls._hip

    def __init__(self, data_or_len, fill=np.nan):
        import ctypes
        a = np.full(int(data_or_len), fill) if np.isscalar(data_or_len) else np.ascontiguousarray(data_or_len, dtype=np.float64)
        self.n = a.size
        p = ctypes.c_void_p()
        assert self.hip().hipMalloc(ctypes.byref(p), max(8 * self.n, 8)) == 0
        self.ptr = p.value
        if self.n:
            assert self.hip().hipMemcpy(self.ptr, a.ctypes.data, 8 * self.n, 1) == 0          # hipMemcpyHostToDevice

    def get(self):
        out = np.empty(self.n)
        self.hip().hipDeviceSynchronize()
        if self.n:
            assert self.hip().hipMemcpy(out.ctypes.data, self.ptr, 8 * self.n, 2) == 0        # hipMemcpyDeviceToHost
        return out

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self.hip().hipFree(self.ptr); self.ptr = None
        except Exception:
            pass


def check_retry_run(timers, status, z, zopt):
    """Outcome of an IPM run on the LP engineered to fail numerically (lp_generators.bump_lp).  WHICH pivot fails, and in which iteration, is
    decided by rounding, and the iterates after a bump depend on it: the same run with another 64 x 64 block kernel (round 5) fails in the same
    matrices (checked matrix by matrix, profiles/r05_bump_trajectories.txt) but walks another trajectory.  Invariant: the retry loop fired, and
    the run ended Trm_Optimal at the HiGHS optimum -- or, the reference's own rule (HSD/step.jl:51, MPC/step.jl:56), Trm_NumericalProblem in a
    step that needed three bumps, by then within 1e-3 of the optimum."""
    assert timers["n_bump"] > 0
    if status == "Trm_Optimal":
        assert abs(z - zopt) <= 1e-6 * (1 + abs(zopt))
    else:
        assert status == "Trm_NumericalProblem" and timers["max_bumps_in_a_step"] >= 3, (status, dict(timers))
        assert abs(z - zopt) <= 1e-3 * (1 + abs(zopt))
