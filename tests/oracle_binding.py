"""ctypes binding of oracle/libk1oracle.so -- the CPU checker (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

OK, NOT_POSDEF, BADARG, NOMEM = 0, 1, 2, 3


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ORACLE_DIR, "libk1oracle.so")
        if not os.path.exists(path):
            build_oracle()
        lib = C.CDLL(path)
        p64, pd, vp = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p
        lib.k1o_setup.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, p64, p64, pd, C.c_int, p64]
        lib.k1o_setup.restype = C.c_int
        lib.k1o_update.argtypes = [vp, pd, pd, pd]
        lib.k1o_update.restype = C.c_int
        lib.k1o_solve.argtypes = [vp, pd, pd, pd, pd]
        lib.k1o_solve.restype = C.c_int
        lib.k1o_free.argtypes = [vp]
        lib.k1o_free.restype = None
        for name in ("k1o_nnzS", "k1o_nnzL", "k1o_fail_col"):
            getattr(lib, name).argtypes = [vp]
            getattr(lib, name).restype = C.c_int64
        lib.k1o_flops.argtypes = [vp]
        lib.k1o_flops.restype = C.c_double
        lib.k1o_get_S.argtypes = [vp, p64, p64, pd]
        lib.k1o_get_L.argtypes = [vp, p64, p64, pd]
        _LIB = lib
    return _LIB


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OraclePosDefError(ArithmeticError):
    pass


class OracleK1:
    """CPU oracle for the K1 path.  A is a scipy.sparse CSC matrix (or anything with
    .indptr/.indices/.data/.shape in CSC layout).  perm: optional permutation, perm[new]=old."""

    def __init__(self, A, perm=None):
        lib = _lib()
        self.m, self.n = A.shape
        self._colptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        self._rowval = np.ascontiguousarray(A.indices, dtype=np.int64)
        self._nzval = np.ascontiguousarray(A.data, dtype=np.float64)
        self._h = C.c_void_p()
        pp = None
        if perm is not None:
            self._perm = np.ascontiguousarray(perm, dtype=np.int64)
            pp = _p64(self._perm)
        rc = lib.k1o_setup(C.byref(self._h), self.m, self.n, _p64(self._colptr), _p64(self._rowval),
                           _pd(self._nzval), 0, pp)
        if rc != OK:
            raise RuntimeError(f"k1o_setup failed rc={rc}")

    def update(self, theta, regP, regD):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        regP = np.ascontiguousarray(regP, dtype=np.float64)
        regD = np.ascontiguousarray(regD, dtype=np.float64)
        assert theta.shape == (self.n,) and regP.shape == (self.n,) and regD.shape == (self.m,)
        rc = _lib().k1o_update(self._h, _pd(theta), _pd(regP), _pd(regD))
        if rc == NOT_POSDEF:
            raise OraclePosDefError(int(_lib().k1o_fail_col(self._h)))
        if rc != OK:
            raise RuntimeError(f"k1o_update rc={rc}")

    def solve(self, xi_p, xi_d):
        xi_p = np.ascontiguousarray(xi_p, dtype=np.float64)
        xi_d = np.ascontiguousarray(xi_d, dtype=np.float64)
        dx = np.empty(self.n)
        dy = np.empty(self.m)
        rc = _lib().k1o_solve(self._h, _pd(dx), _pd(dy), _pd(xi_p), _pd(xi_d))
        if rc != OK:
            raise RuntimeError(f"k1o_solve rc={rc}")
        return dx, dy

    @property
    def nnzS(self):
        return int(_lib().k1o_nnzS(self._h))

    @property
    def nnzL(self):
        return int(_lib().k1o_nnzL(self._h))

    @property
    def flops(self):
        return float(_lib().k1o_flops(self._h))

    def get_S(self):
        import scipy.sparse as sp
        nz = self.nnzS
        cp = np.empty(self.m + 1, dtype=np.int64)
        ri = np.empty(nz, dtype=np.int64)
        v = np.empty(nz)
        _lib().k1o_get_S(self._h, _p64(cp), _p64(ri), _pd(v))
        return sp.csc_matrix((v, ri, cp), shape=(self.m, self.m))

    def get_L(self):
        import scipy.sparse as sp
        nz = self.nnzL
        cp = np.empty(self.m + 1, dtype=np.int64)
        ri = np.empty(nz, dtype=np.int64)
        v = np.empty(nz)
        _lib().k1o_get_L(self._h, _p64(cp), _p64(ri), _pd(v))
        return sp.csc_matrix((v, ri, cp), shape=(self.m, self.m))

    def __del__(self):
        try:
            if self._h:
                _lib().k1o_free(self._h)
                self._h = None
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# K2 (augmented system) oracle: oracle/k2_oracle.c -- pins the next row of the contract
# ------------------------------------------------------------------------------------------------
_LIB2 = None


def _lib2():
    global _LIB2
    if _LIB2 is None:
        path = os.path.join(_ORACLE_DIR, "libk2oracle.so")
        if not os.path.exists(path):
            build_oracle()
        lib = C.CDLL(path)
        p64, pd, vp = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p
        lib.k2o_setup.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, p64, p64, pd, C.c_int, p64]
        lib.k2o_setup.restype = C.c_int
        lib.k2o_update.argtypes = [vp, pd, pd, pd]
        lib.k2o_update.restype = C.c_int
        lib.k2o_solve.argtypes = [vp, pd, pd, pd, pd]
        lib.k2o_solve.restype = C.c_int
        lib.k2o_free.argtypes = [vp]
        lib.k2o_free.restype = None
        for name in ("k2o_nnzK", "k2o_nnzL", "k2o_fail_col"):
            getattr(lib, name).argtypes = [vp]
            getattr(lib, name).restype = C.c_int64
        lib.k2o_get_D.argtypes = [vp, pd]
        _LIB2 = lib
    return _LIB2


class OracleZeroPivotError(ArithmeticError):
    pass


class OracleK2:
    """CPU oracle for the K2 path: LDL' of [-(theta+regP) A'; A regD].  perm: optional permutation of the
    n + m nodes (variables 0..n-1, constraints n..n+m-1), perm[new] = old."""

    def __init__(self, A, perm=None):
        lib = _lib2()
        self.m, self.n = A.shape
        self._colptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        self._rowval = np.ascontiguousarray(A.indices, dtype=np.int64)
        self._nzval = np.ascontiguousarray(A.data, dtype=np.float64)
        self._h = C.c_void_p()
        pp = None
        if perm is not None:
            self._perm = np.ascontiguousarray(perm, dtype=np.int64)
            pp = _p64(self._perm)
        rc = lib.k2o_setup(C.byref(self._h), self.m, self.n, _p64(self._colptr), _p64(self._rowval),
                           _pd(self._nzval), 0, pp)
        if rc != OK:
            raise RuntimeError(f"k2o_setup failed rc={rc}")

    def update(self, theta, regP, regD):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        regP = np.ascontiguousarray(regP, dtype=np.float64)
        regD = np.ascontiguousarray(regD, dtype=np.float64)
        assert theta.shape == (self.n,) and regP.shape == (self.n,) and regD.shape == (self.m,)
        rc = _lib2().k2o_update(self._h, _pd(theta), _pd(regP), _pd(regD))
        if rc == 1:
            raise OracleZeroPivotError(int(_lib2().k2o_fail_col(self._h)))
        if rc != OK:
            raise RuntimeError(f"k2o_update rc={rc}")

    def solve(self, xi_p, xi_d):
        xi_p = np.ascontiguousarray(xi_p, dtype=np.float64)
        xi_d = np.ascontiguousarray(xi_d, dtype=np.float64)
        dx = np.empty(self.n); dy = np.empty(self.m)
        rc = _lib2().k2o_solve(self._h, _pd(dx), _pd(dy), _pd(xi_p), _pd(xi_d))
        if rc != OK:
            raise RuntimeError(f"k2o_solve rc={rc}")
        return dx, dy

    @property
    def nnzL(self):
        return int(_lib2().k2o_nnzL(self._h))

    def D(self):
        d = np.empty(self.n + self.m)
        _lib2().k2o_get_D(self._h, _pd(d))
        return d

    def __del__(self):
        if getattr(self, "_h", None):
            _lib2().k2o_free(self._h)
            self._h = None


# ------------------------------------------------------------------------------------------------
# CPU supernodal comparator: oracle/k1_supernodal.c (CHOLMOD-class speed baseline + second checker)
# ------------------------------------------------------------------------------------------------
_LIB3 = None


def openblas_path():
    """The OpenBLAS shared library that ships inside the SciPy wheel (symbols prefixed `scipy_`)."""
    import glob
    import scipy
    base = os.path.dirname(os.path.dirname(os.path.abspath(scipy.__file__)))
    hits = sorted(glob.glob(os.path.join(base, "scipy.libs", "libscipy_openblas*.so")))
    if not hits:
        raise RuntimeError("SciPy's bundled OpenBLAS not found (scipy.libs/libscipy_openblas*.so)")
    return hits[0]


def _lib3():
    global _LIB3
    if _LIB3 is None:
        path = os.path.join(_ORACLE_DIR, "libk1sn.so")
        if not os.path.exists(path):
            build_oracle()
        lib = C.CDLL(path)
        p64, pd, vp = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p
        i64 = C.c_int64
        lib.k1sn_create.argtypes = ([C.POINTER(vp), C.c_char_p, i64, i64, p64, p64, pd, p64, i64] + [p64] * 10 +
                                    [i64, p64, i64, p64, i64, p64, i64, p64, p64, p64, p64, pd, i64, C.c_int])
        lib.k1sn_create.restype = C.c_int
        lib.k1sn_update.argtypes = [vp, pd, pd, pd]; lib.k1sn_update.restype = C.c_int
        lib.k1sn_solve.argtypes = [vp, pd, pd, pd, pd]; lib.k1sn_solve.restype = C.c_int
        lib.k1sn_free.argtypes = [vp]; lib.k1sn_free.restype = None
        lib.k1sn_fail_col.argtypes = [vp]; lib.k1sn_fail_col.restype = C.c_int64
        lib.k1sn_threads.argtypes = [vp]; lib.k1sn_threads.restype = C.c_int
        lib.k1sn_times.argtypes = [vp, pd]; lib.k1sn_times.restype = None
        lib.k1sn_get_factor.argtypes = [vp, pd, i64]; lib.k1sn_get_factor.restype = C.c_int
        lib.k1sn_get_diag.argtypes = [vp, pd, i64]; lib.k1sn_get_diag.restype = C.c_int
        _LIB3 = lib
    return _LIB3


def _pk_off(lda, col):
    b = col >> 6
    return col * lda - 64 * b * (col - 32 * b - 31)


def _unpacked_targets(s_target, loff_p, loff_u, lda, ns):
    """Assembly targets of the device layout (packed panels) -> offsets into unpacked panels at loff_u."""
    t = np.asarray(s_target, dtype=np.int64)
    out = np.full_like(t, -1)
    live = np.nonzero(t >= 0)[0]
    if live.size == 0:
        return out
    fronts = np.nonzero(loff_p >= 0)[0]
    starts = loff_p[fronts]
    order = np.argsort(starts, kind="stable")
    fronts, starts = fronts[order], starts[order]
    fs = fronts[np.searchsorted(starts, t[live], side="right") - 1]          # front of every entry
    delta = t[live] - loff_p[fs]
    L = lda[fs]
    # slice b: the largest b with S_b <= delta, S_b = 64 b L - 2048 b (b - 1); at most ns / 64 slices -- a short loop over b
    b = np.zeros_like(delta)
    bmax = int((ns[fronts].max() + 63) // 64) if fronts.size else 1
    for cand in range(1, bmax):
        Sb = 64 * cand * L - 2048 * cand * (cand - 1)
        b = np.where((delta >= Sb) & (64 * cand < ns[fs]), cand, b)
    Sb = 64 * b * L - 2048 * b * (b - 1)
    ld = L - 64 * b
    col = 64 * b + (delta - Sb) // ld
    row = 64 * b + (delta - Sb) % ld
    out[live] = loff_u[fs] + row + col * L
    return out


class SupernodalK1:
    """CPU supernodal multifrontal Cholesky (OpenBLAS + OpenMP) on the symbolic structure of a libtlpk
    handle `kkt` (an analyse-only handle is enough): same ordering, same supernodes, same panel
    layout as the device factor.  threads = 0: all cores."""

    def __init__(self, A, kkt, threads=0):
        lib = _lib3()
        from tulip_jl_amd import _lib as tl
        A = A.tocsc(); A.sort_indices()
        self.m, self.n = A.shape
        g = kkt.symbolic
        c64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)   # noqa: E731
        Ap, Ai, Ax = c64(A.indptr), c64(A.indices), np.ascontiguousarray(A.data, dtype=np.float64)
        perm = c64(g("perm"))
        fr = [c64(g(k)) for k in ("front_f", "front_ns", "front_col0", "front_loff", "front_rowoff", "front_reloff",
                                   "front_child_ptr", "front_nchild", "depth", "front_lda")]
        rowidx, rel, children = c64(g("rowidx")), c64(g("rel")), c64(g("children"))
        s_target, s_diag, pair_ptr, pair_j = c64(g("s_target")), c64(g("s_diag_row")), c64(g("pair_ptr")), c64(g("pair_j"))
        pair_w = np.ascontiguousarray(tl.symbolic_array_f64(kkt._h, "pair_w"))
        if pair_w.size != pair_ptr[-1]:
            raise RuntimeError("the handle has released its host assembly lists (device handles do): pass an analyse-only handle")
        # The device stores a panel by 64-column slices (tlpk_host.hpp: pk_off).  The comparator runs dense BLAS on whole panels and
        # keeps them UNPACKED (f x ns, one leading dimension lda): its own offsets, the assembly targets translated; factor_panels()
        # re-packs into the device layout, so that the two factors compare index by index.
        f_, ns_, loff_p, lda_ = fr[0], fr[1], fr[3], fr[9]
        own = loff_p >= 0
        sizes = np.where(own, lda_ * ns_, 0)
        loff_u = np.where(own, np.concatenate([[0], np.cumsum(sizes)[:-1]]), -1).astype(np.int64)
        self._pack = (f_.copy(), ns_.copy(), loff_p.copy(), lda_.copy(), loff_u.copy())
        self.lval_len_packed = int(kkt.stats()["nnzL_stored"])
        self.lval_len = int(sizes.sum())
        s_target = _unpacked_targets(s_target, loff_p, loff_u, lda_, ns_)
        fr[3] = c64(loff_u)
        self._h = C.c_void_p()
        rc = lib.k1sn_create(C.byref(self._h), openblas_path().encode(), self.m, self.n, _p64(Ap), _p64(Ai), _pd(Ax), _p64(perm),
                             len(fr[0]), *[_p64(a) for a in fr], rowidx.size, _p64(rowidx), rel.size, _p64(rel),
                             children.size, _p64(children), s_target.size, _p64(s_target), _p64(s_diag), _p64(pair_ptr),
                             _p64(pair_j), _pd(pair_w), self.lval_len, int(threads))
        if rc != OK:
            raise RuntimeError(f"k1sn_create rc={rc}")
        self.threads = int(lib.k1sn_threads(self._h))

    def update(self, theta, regP, regD):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (theta, regP, regD)]
        rc = _lib3().k1sn_update(self._h, _pd(a[0]), _pd(a[1]), _pd(a[2]))
        if rc == NOT_POSDEF:
            raise OraclePosDefError(int(_lib3().k1sn_fail_col(self._h)))
        if rc != OK:
            raise RuntimeError(f"k1sn_update rc={rc}")

    def solve(self, xi_p, xi_d):
        xp = np.ascontiguousarray(xi_p, dtype=np.float64); xd = np.ascontiguousarray(xi_d, dtype=np.float64)
        dx = np.empty(self.n); dy = np.empty(self.m)
        rc = _lib3().k1sn_solve(self._h, _pd(dx), _pd(dy), _pd(xp), _pd(xd))
        if rc != OK:
            raise RuntimeError(f"k1sn_solve rc={rc}")
        return dx, dy

    def times(self):
        t = np.zeros(3)
        _lib3().k1sn_times(self._h, _pd(t))
        return {"assemble_s": t[0], "factor_s": t[1], "solve_s": t[2]}

    def factor_panels(self):
        """The factor in the DEVICE's panel layout (64-column slices), entries the device never stores dropped."""
        buf = np.empty(max(self.lval_len, 1))
        rc = _lib3().k1sn_get_factor(self._h, _pd(buf), buf.size)
        if rc != OK:
            raise RuntimeError(f"k1sn_get_factor rc={rc}")
        f_, ns_, loff_p, lda_, loff_u = self._pack
        out = np.zeros(max(self.lval_len_packed, 1))
        for s in np.nonzero(loff_p >= 0)[0]:
            f, ns, lda = int(f_[s]), int(ns_[s]), int(lda_[s])
            P = buf[loff_u[s]: loff_u[s] + lda * ns].reshape((lda, ns), order="F")
            for b in range((ns + 63) // 64):
                w = min(64, ns - 64 * b); ld = lda - 64 * b
                start = int(loff_p[s]) + _pk_off(lda, 64 * b) + 64 * b
                out[start: start + w * ld] = P[64 * b:, 64 * b: 64 * b + w].reshape(-1, order="F")
        return out[: self.lval_len_packed]

    def diag(self):
        """diag(L) in permuted order; L_jj^2 is the pivot of column j."""
        d = np.empty(max(self.m, 1))
        rc = _lib3().k1sn_get_diag(self._h, _pd(d), d.size)
        if rc != OK:
            raise RuntimeError(f"k1sn_get_diag rc={rc}")
        return d[: self.m]

    def close(self):
        if getattr(self, "_h", None):
            _lib3().k1sn_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
