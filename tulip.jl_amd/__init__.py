"""tulip.jl_amd -- MI355X-native KKT backend for Tulip's normal-equations Newton step.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels (gfx950) + the C-ABI library (include/tlpk.h)
  _lib.py    ctypes binding of libtlpk.so
  kkt.py     host-side mirror of Tulip's KKT interface (setup / update! / solve!)
  julia/     the Julia glue a Tulip maintainer adds (HIPNormalEquations <: AbstractKKTSolver)
"""
from . import _lib  # noqa: F401
from .kkt import (K1, Backend, DimensionMismatch, HIPNormalEquations, OutOfMemoryError,  # noqa: F401
                  PosDefException, arithmetic, backend, linear_system, run_ls_tests, setup,
                  solve, update)
