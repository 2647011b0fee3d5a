"""ctypes binding of libtlpk.so (C ABI declared in include/tlpk.h).

The library is built in-tree by `make -C tulip.jl_amd/csrc` (see __graft_entry__.build).  There
is no fallback of any kind: if the shared library is missing this module raises, and numeric
calls on a machine without a GPU return TLPK_NO_DEVICE.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TLPK_LIB") or os.path.join(HERE, "libtlpk.so")      # TLPK_LIB: another build of the SAME source tree (A/B of kernel builds)

OK, NOT_POSDEF, BADARG, OOM, HIPERR, NO_DEVICE, TOO_LARGE, NOT_FACTORED, INTERNAL = range(9)
ORDER_AMD, ORDER_NATURAL, ORDER_USER = 0, 1, 2
SYSTEM_K1, SYSTEM_K2 = 0, 1
KC_NAMES = ["assemble", "extend_add", "potrf", "trsm", "update", "solve_fwd", "solve_bwd", "spmv", "update_reduce", "chain"]

p64 = C.POINTER(C.c_int64)
pd = C.POINTER(C.c_double)


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("ordering", C.c_int32),
                ("relax", C.c_int32), ("profile", C.c_int32), ("rank", C.c_int32),
                ("nranks", C.c_int32), ("streams", C.c_int32),
                ("user_perm", p64), ("row_block", p64), ("mem_budget_bytes", C.c_int64),
                ("system", C.c_int32), ("refine_steps", C.c_int32),
                ("detect_blocks", C.c_int32), ("keep_on_too_large", C.c_int32), ("max_link_rows", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("nnzA", C.c_int64), ("nnzS", C.c_int64),
                ("nnzL", C.c_int64), ("nnzL_stored", C.c_int64), ("flops_chol", C.c_double),
                ("flops_panel", C.c_double), ("n_supernodes", C.c_int64), ("n_levels", C.c_int64),
                ("max_front", C.c_int64), ("n_pairs", C.c_int64), ("device_bytes", C.c_int64),
                ("launches_update", C.c_int64), ("launches_solve", C.c_int64),
                ("fail_col", C.c_int64), ("ms_analyse", C.c_double), ("ms_last_update", C.c_double),
                ("ms_last_solve", C.c_double), ("n_local_blocks", C.c_int32), ("n_blocks", C.c_int32),
                ("root_panel_len", C.c_int64), ("flops_update", C.c_double),
                ("flops_update_alg", C.c_double), ("ms_enqueue_update", C.c_double), ("refine_rejected", C.c_int64),
                ("flops_update_chain", C.c_double), ("flops_update_alg_chain", C.c_double), ("chain_launches", C.c_int64), ("chain_items", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_double * len(KC_NAMES)), ("launches", C.c_int64 * len(KC_NAMES))]

    def as_dict(self):
        return {KC_NAMES[i]: {"ms": self.ms[i], "launches": self.launches[i]} for i in range(len(KC_NAMES))}


_lib = None

EXPORTS = [
    "tlpk_default_options", "tlpk_create", "tlpk_destroy", "tlpk_update", "tlpk_solve",
    "tlpk_update_device", "tlpk_solve_device", "tlpk_sync", "tlpk_stream", "tlpk_update_local",
    "tlpk_root_panel", "tlpk_update_finish", "tlpk_solve_local", "tlpk_root_rhs",
    "tlpk_solve_finish", "tlpk_info", "tlpk_kernel_timing", "tlpk_get_perm", "tlpk_symbolic_get",
    "tlpk_symbolic_get_f64", "tlpk_set_profile", "tlpk_root_copy", "tlpk_get_factor", "tlpk_strerror", "tlpk_last_error",
    "tlpk_backend_name", "tlpk_system_name", "tlpk_linear_system", "tlpk_device_count",
    "tlpk_create_multi", "tlpk_ipm_load", "tlpk_ipm_reset", "tlpk_ipm_residuals", "tlpk_ipm_factor", "tlpk_ipm_hsolve", "tlpk_ipm_targets",
    "tlpk_ipm_newton", "tlpk_ipm_accept", "tlpk_ipm_advance", "tlpk_ipm_get",
    "tlpk_mpc_start", "tlpk_mpc_newton", "tlpk_mpc_gap", "tlpk_mpc_targets", "tlpk_mpc_advance",
    "tlpk_detect_blocks", "tlpk_solve2_device", "tlpk_ipm_hsolve_newton", "tlpk_update_device_async", "tlpk_ipm_factor_hsolve_newton",
    "tlpk_refine_local", "tlpk_refine_finish", "tlpk_solve2_local", "tlpk_root_rhs2", "tlpk_solve2_finish", "tlpk_last_create_error",
    "tlpk_host_copy_threads",
]


def lib():
    """Load libtlpk.so and declare the prototypes of include/tlpk.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C tulip.jl_amd/csrc)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.tlpk_default_options.argtypes = [C.POINTER(Options)]
    L.tlpk_default_options.restype = None
    L.tlpk_create.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, p64, p64, pd, C.c_int, C.POINTER(Options)]
    L.tlpk_create_multi.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, p64, p64, pd, C.c_int, C.POINTER(Options), C.c_int, C.POINTER(C.c_int32)]
    L.tlpk_create_multi.restype = C.c_int
    L.tlpk_destroy.argtypes = [vp]
    L.tlpk_destroy.restype = None
    L.tlpk_update.argtypes = [vp, pd, pd, pd]
    L.tlpk_solve.argtypes = [vp, pd, pd, pd, pd]
    L.tlpk_update_device.argtypes = [vp, vp, vp, vp]
    L.tlpk_solve_device.argtypes = [vp, vp, vp, vp, vp]
    L.tlpk_update_device_async.argtypes = [vp, vp, vp, vp]
    L.tlpk_update_device_async.restype = C.c_int
    L.tlpk_solve2_device.argtypes = [vp] + [vp] * 8
    L.tlpk_solve2_device.restype = C.c_int
    L.tlpk_sync.argtypes = [vp]
    L.tlpk_stream.argtypes = [vp]
    L.tlpk_stream.restype = vp
    L.tlpk_update_local.argtypes = [vp, vp, vp, vp]
    L.tlpk_root_panel.argtypes = [vp, C.POINTER(vp), p64]
    L.tlpk_update_finish.argtypes = [vp]
    L.tlpk_solve_local.argtypes = [vp, vp, vp]
    L.tlpk_root_rhs.argtypes = [vp, C.POINTER(vp), p64]
    L.tlpk_solve_finish.argtypes = [vp, vp, vp, vp]
    L.tlpk_refine_local.argtypes = [vp, vp, vp, vp, vp]
    L.tlpk_refine_local.restype = C.c_int
    L.tlpk_refine_finish.argtypes = [vp, vp, vp]
    L.tlpk_refine_finish.restype = C.c_int
    L.tlpk_solve2_local.argtypes = [vp, vp, vp, vp, vp]
    L.tlpk_solve2_local.restype = C.c_int
    L.tlpk_root_rhs2.argtypes = [vp, C.POINTER(vp), p64]
    L.tlpk_root_rhs2.restype = C.c_int
    L.tlpk_solve2_finish.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.tlpk_solve2_finish.restype = C.c_int
    L.tlpk_root_copy.argtypes = [vp, C.c_int, C.c_int, vp]
    L.tlpk_root_copy.restype = C.c_int
    L.tlpk_info.argtypes = [vp, C.POINTER(Stats)]
    L.tlpk_kernel_timing.argtypes = [vp, C.POINTER(KernelTimes)]
    L.tlpk_set_profile.argtypes = [vp, C.c_int]
    L.tlpk_set_profile.restype = C.c_int
    L.tlpk_get_perm.argtypes = [vp, p64]
    L.tlpk_symbolic_get.argtypes = [vp, C.c_char_p, p64, C.c_int64]
    L.tlpk_symbolic_get.restype = C.c_int64
    L.tlpk_symbolic_get_f64.argtypes = [vp, C.c_char_p, pd, C.c_int64]
    L.tlpk_symbolic_get_f64.restype = C.c_int64
    L.tlpk_get_factor.argtypes = [vp, pd, C.c_int64]
    L.tlpk_strerror.argtypes = [C.c_int]
    L.tlpk_strerror.restype = C.c_char_p
    L.tlpk_last_error.argtypes = [vp]
    L.tlpk_last_error.restype = C.c_char_p
    L.tlpk_last_create_error.argtypes = []
    L.tlpk_last_create_error.restype = C.c_char_p
    L.tlpk_backend_name.restype = C.c_char_p
    L.tlpk_system_name.restype = C.c_char_p
    L.tlpk_linear_system.argtypes = [vp]
    L.tlpk_linear_system.restype = C.c_char_p
    L.tlpk_device_count.restype = C.c_int
    L.tlpk_host_copy_threads.restype = C.c_int
    L.tlpk_detect_blocks.argtypes = [C.c_int64, C.c_int64, p64, p64, C.c_int, C.c_int64, p64, p64, p64]
    L.tlpk_detect_blocks.restype = C.c_int
    L.tlpk_ipm_load.argtypes = [vp, pd, pd, pd, pd]
    L.tlpk_ipm_reset.argtypes = [vp]
    L.tlpk_ipm_residuals.argtypes = [vp, C.c_double, pd]
    L.tlpk_ipm_factor.argtypes = [vp, C.c_double, C.c_double]
    L.tlpk_ipm_hsolve.argtypes = [vp, pd]
    L.tlpk_ipm_targets.argtypes = [vp, C.c_double, C.c_double, C.c_double, pd]
    L.tlpk_ipm_newton.argtypes = [vp, C.c_int, pd, pd]
    L.tlpk_ipm_factor_hsolve_newton.argtypes = [vp, C.c_double, C.c_double, pd, pd]
    L.tlpk_ipm_factor_hsolve_newton.restype = C.c_int
    L.tlpk_ipm_hsolve_newton.argtypes = [vp, pd, pd]
    L.tlpk_ipm_hsolve_newton.restype = C.c_int
    L.tlpk_ipm_accept.argtypes = [vp]
    L.tlpk_ipm_advance.argtypes = [vp, C.c_double, pd]
    L.tlpk_ipm_get.argtypes = [vp, C.c_int, pd, C.c_int64]
    L.tlpk_mpc_start.argtypes = [vp, pd]
    L.tlpk_mpc_newton.argtypes = [vp, C.c_int, C.c_double, pd]
    L.tlpk_mpc_gap.argtypes = [vp, C.c_double, C.c_double, pd]
    L.tlpk_mpc_targets.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double]
    L.tlpk_mpc_advance.argtypes = [vp, C.c_double, C.c_double, pd]
    for name in ("tlpk_ipm_load", "tlpk_ipm_reset", "tlpk_ipm_residuals", "tlpk_ipm_factor", "tlpk_ipm_hsolve",
                 "tlpk_ipm_targets", "tlpk_ipm_newton", "tlpk_ipm_accept", "tlpk_ipm_advance", "tlpk_ipm_get",
                 "tlpk_mpc_start", "tlpk_mpc_newton", "tlpk_mpc_gap", "tlpk_mpc_targets", "tlpk_mpc_advance"):
        getattr(L, name).restype = C.c_int
    for name in ("tlpk_create", "tlpk_update", "tlpk_solve", "tlpk_update_device", "tlpk_solve_device",
                 "tlpk_sync", "tlpk_update_local", "tlpk_root_panel", "tlpk_update_finish",
                 "tlpk_solve_local", "tlpk_root_rhs", "tlpk_solve_finish", "tlpk_info",
                 "tlpk_kernel_timing", "tlpk_get_perm", "tlpk_get_factor"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def as_p64(a):
    return a.ctypes.data_as(p64)


def as_pd(a):
    return a.ctypes.data_as(pd)


def strerror(code):
    return lib().tlpk_strerror(code).decode()


def symbolic_array(handle, what):
    """Fetch one of the symbolic arrays (tests and tools)."""
    L = lib()
    n = L.tlpk_symbolic_get(handle, what.encode(), None, 0)
    if n < 0:
        raise KeyError(what)
    buf = np.empty(max(n, 1), dtype=np.int64)
    L.tlpk_symbolic_get(handle, what.encode(), as_p64(buf), n)
    return buf[:n]


def symbolic_array_f64(handle, what):
    L = lib()
    n = L.tlpk_symbolic_get_f64(handle, what.encode(), None, 0)
    if n < 0:
        raise KeyError(what)
    buf = np.empty(max(n, 1), dtype=np.float64)
    L.tlpk_symbolic_get_f64(handle, what.encode(), as_pd(buf), n)
    return buf[:n]
