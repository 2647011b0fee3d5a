"""tulip.jl_amd -- MI355X-native KKT backend for Tulip's normal-equations Newton step.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels (gfx950) + the C-ABI library (include/tlpk.h)
  _lib.py    ctypes binding of libtlpk.so
  kkt.py     host-side mirror of Tulip's KKT interface (setup / update! / solve!)
  hsd_device.py / mpc_device.py  optional: Tulip's HSD and MPC loops with the iterate resident in HBM (tlpk_ipm_* / tlpk_mpc_*), scalars only over PCIe
  problem.py / presolve.py / model.py  front end: free-MPS reader, standard form, presolve + scaling + postsolve, Model
  julia/     the Julia glue a Tulip maintainer adds (HIPNormalEquations <: AbstractKKTSolver)
"""
import os as _os

# The library runs up to 4 HIP streams concurrently (2 stream groups x (stream + side stream)).  The
# ROCm runtime multiplexes all streams of the process onto GPU_MAX_HW_QUEUES hardware queues (4 by
# default); when the host application owns streams too, two of ours can land on one queue and
# serialise (measured on config C4: 69.7 vs 76.6-80.2 ms per Newton step).  The variable is read
# when the HIP runtime initialises, i.e. at the first HIP call of the process, so setting it here
# works as long as this package is imported before that.  A tuning knob of the HOST process: libtlpk.so
# itself never touches the environment (the Julia shim julia/libtlpk.jl sets it the same way).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: F401,E402
from .kkt import (K1, K2, Backend, DimensionMismatch, HIPNormalEquations, OutOfMemoryError,  # noqa: F401,E402
                  PosDefException, arithmetic, backend, linear_system, run_ls_tests, setup,
                  solve, update)
from .model import Model  # noqa: F401,E402
from .presolve import Presolve, PresolveOptions  # noqa: F401,E402
from .problem import LP, read_free_mps, standard_form  # noqa: F401,E402
