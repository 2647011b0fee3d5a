"""Host-side mirror of Tulip's KKT interface for the HIP normal-equations backend.

Same names, argument meaning and error behaviour as the reference
(/root/reference/src/KKT/KKT.jl:59-121, src/KKT/Cholmod/spd.jl:5-70); Julia's `update!` /
`solve!` are `update` / `solve` here.  The Julia glue with the same semantics is
tulip.jl_amd/julia/hip.jl.  All numeric work happens in libtlpk.so on the GPU; this module only
validates arguments and maps return codes to exceptions.
"""
import ctypes as C

import numpy as np

from . import _lib

SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))


class K1:
    """Normal-equations system, /root/reference/src/KKT/systems.jl:34-54."""


class K2:
    """Augmented system, /root/reference/src/KKT/systems.jl:12-31 -- the reference's default for Float64
    (KKT.jl:134-141, Cholmod/sqd.jl).  On the device: signed Cholesky P K P' = L S L' of the quasi-definite
    matrix [-(Theta^-1 + Rp) A'; A Rd]."""


class DimensionMismatch(ValueError):
    """Julia's DimensionMismatch (spd.jl:26-34)."""


class PosDefException(ArithmeticError):
    """Julia's PosDefException (spd.jl:47); caught by the IPM to bump regularisations
    (/root/reference/src/IPM/HSD/step.jl:39-48)."""

    def __init__(self, info=0):
        super().__init__(f"matrix is not positive definite; Cholesky factorization failed (column {info})")
        self.info = info


class OutOfMemoryError(MemoryError):
    """Julia's OutOfMemoryError -> Trm_MemoryLimit (/root/reference/src/IPM/HSD/HSD.jl:327-329)."""


class Backend:
    """`TlpHIP.Backend`: selects the HIP solver, the analogue of `TlpCholmod.Backend`
    (/root/reference/src/KKT/Cholmod/cholmod.jl:18).

    row_block: optional block-angular structure (length m; block id >= 0 or -1 for a linking
    row) -- Tulip's structured-matrix hook (parameters.jl:11 MatrixFactory -> KKT.setup dispatch) --
    or "auto": the library finds the structure of the matrix it is given (tlpk_detect_blocks).  An explicit
    map indexes the rows of the matrix KKT.setup receives, i.e. it is only meaningful without presolve
    (Tulip's presolve removes and renumbers rows, model.jl:88-131); "auto" works on the presolved matrix.
    """

    def __init__(self, device=0, ordering="amd", relax=True, row_block=None, user_perm=None,
                 profile=False, rank=0, nranks=1, mem_budget_bytes=0, streams=0, ngpus=1, devices=None, refine=0,
                 max_link_rows=0):
        self.device = device
        self.ordering = {"amd": _lib.ORDER_AMD, "natural": _lib.ORDER_NATURAL, "user": _lib.ORDER_USER}[ordering]
        self.relax = bool(relax)
        self.detect_blocks = isinstance(row_block, str)
        if self.detect_blocks:
            if row_block != "auto":
                raise ValueError("row_block: a vector of block ids, None, or 'auto'")
            row_block = None
        self.max_link_rows = int(max_link_rows)
        self.row_block = None if row_block is None else np.ascontiguousarray(row_block, dtype=np.int64)
        self.user_perm = None if user_perm is None else np.ascontiguousarray(user_perm, dtype=np.int64)
        self.profile = bool(profile)
        self.rank, self.nranks = int(rank), int(nranks)
        self.mem_budget_bytes = int(mem_budget_bytes)
        self.streams = int(streams)            # 0 = auto (2 concurrent block groups), 1 = single group
        self.refine = int(refine)              # iterative-refinement steps per solve (0 = the reference's behaviour; K1; one rank or ngpus > 1 -- sharded handles: refine_local / refine_finish)
        # single-process multi-GPU (block-angular LPs): one handle shards the diagonal blocks over `ngpus` devices
        self.ngpus = int(ngpus)
        self.devices = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)


def detect_blocks(A, max_link_rows=0):
    """tlpk_detect_blocks on a scipy sparse matrix: (row_block, n_blocks, n_link); n_blocks == 1: no block structure."""
    import scipy.sparse as sp
    A = sp.csc_matrix(A)
    m, n = A.shape
    colptr = np.ascontiguousarray(A.indptr, dtype=np.int64); rowval = np.ascontiguousarray(A.indices, dtype=np.int64)
    rb = np.zeros(max(m, 1), dtype=np.int64)
    nb, nl = C.c_int64(1), C.c_int64(0)
    rc = _lib.lib().tlpk_detect_blocks(m, n, _lib.as_p64(colptr), _lib.as_p64(rowval), 0, int(max_link_rows), _lib.as_p64(rb),
                                       C.byref(nb), C.byref(nl))
    _raise_for(rc, None, "tlpk_detect_blocks: ")
    return rb[:m], int(nb.value), int(nl.value)


def _raise_for(code, handle=None, what=""):
    if code == _lib.OK:
        return
    msg = _lib.strerror(code)
    # a failed create returns no handle: its diagnostic is tlpk_last_create_error()
    detail = _lib.lib().tlpk_last_error(handle).decode() if handle else (_lib.lib().tlpk_last_create_error().decode() if what.startswith("KKT.setup") else "")
    if detail:
        msg = f"{msg}: {detail}"
    if code == _lib.NOT_POSDEF:
        raise PosDefException(0)
    if code == _lib.BADARG:
        raise DimensionMismatch(f"{what}{msg}")
    if code in (_lib.OOM, _lib.TOO_LARGE):
        raise OutOfMemoryError(msg)
    raise RuntimeError(f"{what}{msg} (code {code})")


class HIPNormalEquations:
    """`HIPNormalEquations <: AbstractKKTSolver{Float64}` -- the counterpart of
    `CholmodSolver{Float64,K1}` (/root/reference/src/KKT/Cholmod/cholmod.jl:46-60)."""

    def __init__(self, A, backend_, system=_lib.SYSTEM_K1):
        import scipy.sparse as sp
        if not sp.issparse(A):
            A = sp.csc_matrix(np.asarray(A, dtype=np.float64))     # cholmod.jl:65 convert(SparseMatrixCSC, A)
        A = A.tocsc()
        A.sort_indices()
        self.m, self.n = A.shape
        self.A = A                                                # stored by reference, never mutated
        L = _lib.lib()
        opt = _lib.Options()
        L.tlpk_default_options(C.byref(opt))
        opt.device = backend_.device
        opt.ordering = backend_.ordering
        opt.relax = int(backend_.relax)
        opt.profile = int(backend_.profile)
        opt.rank, opt.nranks = backend_.rank, backend_.nranks
        opt.mem_budget_bytes = backend_.mem_budget_bytes
        opt.streams = backend_.streams
        opt.refine_steps = backend_.refine
        opt.detect_blocks = int(getattr(backend_, "detect_blocks", False))
        opt.max_link_rows = int(getattr(backend_, "max_link_rows", 0))
        opt.system = system
        self.system = system
        self._keep = []
        if backend_.row_block is not None:
            if backend_.row_block.shape != (self.m,):
                raise DimensionMismatch(f"length(row_block)={backend_.row_block.shape[0]} but A has m={self.m}")
            opt.row_block = _lib.as_p64(backend_.row_block)
            self._keep.append(backend_.row_block)
        if backend_.user_perm is not None:
            opt.user_perm = _lib.as_p64(backend_.user_perm)
            self._keep.append(backend_.user_perm)
        colptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(A.indices, dtype=np.int64)
        nzval = np.ascontiguousarray(A.data, dtype=np.float64)
        self._h = C.c_void_p()
        if backend_.ngpus > 1:
            if backend_.devices is not None and backend_.devices.shape != (backend_.ngpus,):
                raise DimensionMismatch("devices must list ngpus ordinals")
            dv = None if backend_.devices is None else backend_.devices.ctypes.data_as(C.POINTER(C.c_int32))
            rc = L.tlpk_create_multi(C.byref(self._h), self.m, self.n, _lib.as_p64(colptr), _lib.as_p64(rowval),
                                     _lib.as_pd(nzval), 0, C.byref(opt), backend_.ngpus, dv)
        else:
            rc = L.tlpk_create(C.byref(self._h), self.m, self.n, _lib.as_p64(colptr), _lib.as_p64(rowval),
                               _lib.as_pd(nzval), 0, C.byref(opt))
        if rc != _lib.OK:
            h = self._h if self._h else None
            try:
                _raise_for(rc, h, "KKT.setup: ")
            finally:
                if self._h:
                    L.tlpk_destroy(self._h)
                    self._h = C.c_void_p()
        self.backend_options = backend_

    # -- introspection --
    def stats(self):
        st = _lib.Stats()
        _lib.lib().tlpk_info(self._h, C.byref(st))
        return st.as_dict()

    def kernel_times(self):
        kt = _lib.KernelTimes()
        _lib.lib().tlpk_kernel_timing(self._h, C.byref(kt))
        return kt.as_dict()

    def set_profile(self, on):
        _lib.lib().tlpk_set_profile(self._h, int(bool(on)))

    # -- split-phase entry points for block-angular sharding (include/tlpk.h) --
    def update_local(self, d_theta, d_regP, d_regD):
        _raise_for(_lib.lib().tlpk_update_local(self._h, d_theta, d_regP, d_regD), self._h)

    def update_finish(self):
        _raise_for(_lib.lib().tlpk_update_finish(self._h), self._h)

    def solve_local(self, d_xip, d_xid):
        _raise_for(_lib.lib().tlpk_solve_local(self._h, d_xip, d_xid), self._h)

    def solve_finish(self, d_dx, d_dy, d_xid):
        _raise_for(_lib.lib().tlpk_solve_finish(self._h, d_dx, d_dy, d_xid), self._h)

    def solve2_local(self, d_xip0, d_xid0, d_xip1, d_xid1):
        """First half of a pair of solves on a sharded handle (then: all-reduce root_rhs() and root_rhs2(), solve2_finish)."""
        _raise_for(_lib.lib().tlpk_solve2_local(self._h, d_xip0, d_xid0, d_xip1, d_xid1), self._h)

    def solve2_finish(self, d_dx0, d_dy0, d_xid0, d_dx1, d_dy1, d_xid1):
        _raise_for(_lib.lib().tlpk_solve2_finish(self._h, d_dx0, d_dy0, d_xid0, d_dx1, d_dy1, d_xid1), self._h)

    def root_rhs2(self):
        """(device address, count) of the root right-hand side of the SECOND system of a pair."""
        p = C.c_void_p(); n = C.c_int64()
        _raise_for(_lib.lib().tlpk_root_rhs2(self._h, C.byref(p), C.byref(n)), self._h)
        return p.value, n.value

    def refine_local(self, d_dx, d_dy, d_xip, d_xid):
        """First half of one iterative-refinement step on a sharded handle (then: all-reduce root_rhs(), refine_finish)."""
        _raise_for(_lib.lib().tlpk_refine_local(self._h, d_dx, d_dy, d_xip, d_xid), self._h)

    def refine_finish(self, d_dx, d_dy):
        _raise_for(_lib.lib().tlpk_refine_finish(self._h, d_dx, d_dy), self._h)

    def root_panel(self):
        """(device address, count) of the root (linking) panel to all-reduce after update_local."""
        p = C.c_void_p(); n = C.c_int64()
        _raise_for(_lib.lib().tlpk_root_panel(self._h, C.byref(p), C.byref(n)), self._h)
        return (p.value or 0), n.value

    def root_copy(self, which, direction, d_buf):
        """which: 'panel' | 'rhs' | 'rhs2' (second system of a pair); direction: 'out' (library -> d_buf) | 'in'; async on the stream."""
        rc = _lib.lib().tlpk_root_copy(self._h, {"panel": 0, "rhs": 1, "rhs2": 2}[which], 0 if direction == "out" else 1, d_buf)
        _raise_for(rc, self._h)

    def root_rhs(self):
        p = C.c_void_p(); n = C.c_int64()
        _raise_for(_lib.lib().tlpk_root_rhs(self._h, C.byref(p), C.byref(n)), self._h)
        return (p.value or 0), n.value

    def perm(self):
        """perm[new] = old.  K1: the m rows of S; K2: the n + m nodes of the augmented matrix (variables 0..n-1,
        constraints n..n+m-1)."""
        p = np.empty(self.m + self.n if self.system == _lib.SYSTEM_K2 else self.m, dtype=np.int64)
        _lib.lib().tlpk_get_perm(self._h, _lib.as_p64(p))
        return p

    def symbolic(self, what):
        return _lib.symbolic_array(self._h, what)

    def factor_panels(self):
        st = self.stats()
        buf = np.empty(max(st["nnzL_stored"], 1))
        _raise_for(_lib.lib().tlpk_get_factor(self._h, _lib.as_pd(buf), buf.size), self._h)
        return buf[:st["nnzL_stored"]]

    # -- device-pointer entry points (arguments: integer device addresses) --
    def update_device(self, d_theta, d_regP, d_regD):
        _raise_for(_lib.lib().tlpk_update_device(self._h, d_theta, d_regP, d_regD), self._h)

    def update_device_async(self, d_theta, d_regP, d_regD):
        """tlpk_update_device_async: no wait; the verdict (PosDefException) comes from the next sync()."""
        _raise_for(_lib.lib().tlpk_update_device_async(self._h, d_theta, d_regP, d_regD), self._h)

    def solve_device(self, d_dx, d_dy, d_xip, d_xid, sync=True):
        _raise_for(_lib.lib().tlpk_solve_device(self._h, d_dx, d_dy, d_xip, d_xid), self._h)
        if sync:
            _raise_for(_lib.lib().tlpk_sync(self._h), self._h)

    def solve2_device(self, d_dx0, d_dy0, d_xip0, d_xid0, d_dx1, d_dy1, d_xip1, d_xid1, sync=True):
        """Two right-hand sides in one pass over the factor (tlpk_solve2_device); bit-identical to two solve_device calls."""
        _raise_for(_lib.lib().tlpk_solve2_device(self._h, d_dx0, d_dy0, d_xip0, d_xid0, d_dx1, d_dy1, d_xip1, d_xid1), self._h)
        if sync:
            _raise_for(_lib.lib().tlpk_sync(self._h), self._h)

    def sync(self):
        _raise_for(_lib.lib().tlpk_sync(self._h), self._h)

    def stream_ptr(self):
        """hipStream_t of the handle (an integer address), e.g. for torch.cuda.ExternalStream."""
        return _lib.lib().tlpk_stream(self._h) or 0

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().tlpk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):                       # the Julia glue does this in a finalizer
        try:
            self.close()
        except Exception:
            pass


def setup(A, system=None, backend_=None):
    """KKT.setup(A, ::K1, ::TlpHIP.Backend)  (KKT.jl:59, spd.jl:5-20).  Runs the analyse phase on
    the host and uploads the symbolic structures; skips the throw-away numeric factorisation of
    spd.jl:14-17."""
    if system is None or isinstance(system, K1) or system is K1:
        return HIPNormalEquations(A, backend_ or Backend())
    if isinstance(system, K2) or system is K2:
        return HIPNormalEquations(A, backend_ or Backend(), system=_lib.SYSTEM_K2)
    raise TypeError("the HIP backend solves the normal equations (K1) or the augmented system (K2)")


def _vec(x, name, length, what):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if a.ndim != 1 or a.shape[0] != length[1]:
        raise DimensionMismatch(f"length({name})={a.shape[0] if a.ndim == 1 else a.shape} "
                                f"but KKT solver has {length[0]}={length[1]}{what}")
    return a


def update(kkt, theta_inv, regP, regD):
    """KKT.update!(kkt, θinv, regP, regD) -> nothing  (KKT.jl:65-83, spd.jl:22-50).
    Raises DimensionMismatch (spd.jl:26-34) or PosDefException (spd.jl:47)."""
    th = _vec(theta_inv, "θ", ("n", kkt.n), ".")
    rp = _vec(regP, "regP", ("n", kkt.n), "")
    rd = _vec(regD, "regD", ("m", kkt.m), "")
    rc = _lib.lib().tlpk_update(kkt._h, _lib.as_pd(th), _lib.as_pd(rp), _lib.as_pd(rd))
    _raise_for(rc, kkt._h, "KKT.update!: ")
    return None


def solve(dx, dy, kkt, xi_p, xi_d):
    """KKT.solve!(dx, dy, kkt, ξp, ξd) -> nothing  (KKT.jl:85-100, spd.jl:52-70).
    dx, dy are overwritten in place; ξp, ξd are read-only."""
    if not (isinstance(dx, np.ndarray) and isinstance(dy, np.ndarray) and dx.dtype == np.float64
            and dy.dtype == np.float64 and dx.flags.c_contiguous and dy.flags.c_contiguous):
        raise TypeError("dx, dy must be contiguous float64 numpy vectors (modified in place)")
    if dx.shape != (kkt.n,):
        raise DimensionMismatch(f"length(dx)={dx.shape[0]} but KKT solver has n={kkt.n}")
    if dy.shape != (kkt.m,):
        raise DimensionMismatch(f"length(dy)={dy.shape[0]} but KKT solver has m={kkt.m}")
    xp = _vec(xi_p, "ξp", ("m", kkt.m), "")
    xd = _vec(xi_d, "ξd", ("n", kkt.n), "")
    rc = _lib.lib().tlpk_solve(kkt._h, _lib.as_pd(dx), _lib.as_pd(dy), _lib.as_pd(xp), _lib.as_pd(xd))
    _raise_for(rc, kkt._h, "KKT.solve!: ")
    return None


def arithmetic(kkt):
    """KKT.arithmetic (KKT.jl:107)."""
    return np.float64


def backend(kkt):
    """KKT.backend (KKT.jl:114) -- printed in the IPM log banner (HSD.jl:227-229)."""
    return _lib.lib().tlpk_backend_name().decode()


def linear_system(kkt):
    """KKT.linear_system (KKT.jl:121; spd.jl:3)."""
    return _lib.lib().tlpk_linear_system(kkt._h if kkt is not None else None).decode()


def run_ls_tests(A, kkt, atol=SQRT_EPS):
    """The reference's conformance routine for a KKT backend, restated
    (/root/reference/src/KKT/Test/test.jl:9-47): update! with all-ones, solve! with all-ones,
    both residual infinity-norms <= atol.  Returns (rp_norm, rd_norm)."""
    import scipy.sparse as sp
    assert callable(update) and callable(solve)          # test.jl:19-20 hasmethod checks
    Ad = A if sp.issparse(A) else sp.csc_matrix(np.asarray(A, dtype=np.float64))
    m, n = Ad.shape
    th = np.ones(n); rp = np.ones(n); rd = np.ones(m)
    update(kkt, th, rp, rd)                              # test.jl:26-29
    xp = np.ones(m); xd = np.ones(n)
    dx = np.zeros(n); dy = np.zeros(m)
    solve(dx, dy, kkt, xp, xd)                           # test.jl:32-36
    r_p = Ad @ dx + rd * dy - xp                         # test.jl:39
    r_d = -dx * (th + rp) + Ad.T @ dy - xd               # test.jl:40
    np_, nd_ = float(np.abs(r_p).max(initial=0.0)), float(np.abs(r_d).max(initial=0.0))
    if not (np_ <= atol and nd_ <= atol):
        raise AssertionError(f"run_ls_tests residuals {np_:.3e}, {nd_:.3e} exceed atol={atol:.3e}")
    return np_, nd_
