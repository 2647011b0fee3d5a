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
