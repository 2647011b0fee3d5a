"""The C-ABI library loads on a CPU-only machine and exports every symbol include/tlpk.h declares
(no compute calls here).  Also: the product package never imports or links the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import tulip_jl_amd as tk
from tulip_jl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "tlpk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(tlpk_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_every_declared_symbol_is_exported():
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"libtlpk.so does not export {n}"
    # and the Python binding knows about all of them
    assert set(names) <= set(_lib.EXPORTS), set(names) - set(_lib.EXPORTS)


def test_struct_layouts_match_header():
    opt = _lib.Options()
    _lib.lib().tlpk_default_options(ctypes.byref(opt))
    assert opt.struct_size == ctypes.sizeof(_lib.Options)        # the library checks this on create
    assert opt.nranks == 1 and opt.relax == 1 and opt.ordering == _lib.ORDER_AMD


def test_strings_and_error_text():
    assert tk.backend(None) == "HIP (gfx950)"
    assert tk.linear_system(None) == "Normal equations (K1)"
    assert "positive definite" in _lib.strerror(_lib.NOT_POSDEF)
    assert _lib.lib().tlpk_device_count() >= 0


def test_product_does_not_reference_the_oracle():
    """No CPU fallback: the product sources never mention the oracle, and libtlpk.so does not
    link it."""
    pkg = os.path.join(ROOT, "tulip.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".jl")):
                src = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "k1o_" not in src and "k2o_" not in src and "libk1oracle" not in src and "libk2oracle" not in src \
                    and "oracle_binding" not in src, fn
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "k1oracle" not in out and "k2oracle" not in out
    assert "amdhip64" in out


def test_null_and_bad_arguments_return_codes():
    L = _lib.lib()
    assert L.tlpk_update(None, None, None, None) == _lib.BADARG
    h = ctypes.c_void_p()
    colptr = np.array([0, 1], dtype=np.int64)
    rowval = np.array([5], dtype=np.int64)              # row index out of range for m = 2
    nz = np.array([1.0])
    opt = _lib.Options(); L.tlpk_default_options(ctypes.byref(opt)); opt.device = -1
    rc = L.tlpk_create(ctypes.byref(h), 2, 1, _lib.as_p64(colptr), _lib.as_p64(rowval), _lib.as_pd(nz), 0, ctypes.byref(opt))
    assert rc == _lib.BADARG and not h
    opt.struct_size = 3
    rc = L.tlpk_create(ctypes.byref(h), 2, 1, _lib.as_p64(colptr), _lib.as_p64(rowval), _lib.as_pd(nz), 0, ctypes.byref(opt))
    assert rc == _lib.BADARG


def test_one_based_indices_like_julia():
    import scipy.sparse as sp
    from helpers import random_lp_matrix
    A = random_lp_matrix(30, 50, 3, 3)
    k0 = tk.setup(A, tk.K1(), tk.Backend(device=-1))
    L = _lib.lib()
    h = ctypes.c_void_p()
    colptr = (A.indptr.astype(np.int64) + 1); rowval = (A.indices.astype(np.int64) + 1)
    opt = _lib.Options(); L.tlpk_default_options(ctypes.byref(opt)); opt.device = -1
    rc = L.tlpk_create(ctypes.byref(h), 30, 50, _lib.as_p64(colptr), _lib.as_p64(rowval), _lib.as_pd(A.data), 1, ctypes.byref(opt))
    assert rc == _lib.OK
    p = np.empty(30, dtype=np.int64)
    L.tlpk_get_perm(h, _lib.as_p64(p))
    assert (p == k0.perm()).all()
    L.tlpk_destroy(h)


def _build_abi_smoke(tmp_path):
    import shutil
    import subprocess
    exe = str(tmp_path / "abi_smoke")
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "no C compiler"
    subprocess.check_call([cc, "-O1", "-Wall", "-Wextra", "-std=c99", "-o", exe, os.path.join(HERE, "abi_smoke.c"), "-ldl", "-lm"])
    return exe


def test_plain_c_caller_one_based_int64_csc_analyse_only(tmp_path):
    """tests/abi_smoke.c = what the Julia `ccall`s do (1-based Int64 CSC, plain pointers): on a machine
    without a GPU every numeric call must return TLPK_NO_DEVICE (no CPU fallback)."""
    import subprocess
    exe = _build_abi_smoke(tmp_path)
    out = subprocess.run([exe, _lib.LIB_PATH, "cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke cpu ok" in out.stdout


@pytest.mark.gpu
def test_plain_c_caller_drives_setup_update_solve_on_device(tmp_path):
    import subprocess
    exe = _build_abi_smoke(tmp_path)
    out = subprocess.run([exe, _lib.LIB_PATH, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke gpu ok" in out.stdout


def test_multi_device_create_rejects_what_it_cannot_do():
    """tlpk_create_multi needs a block-angular K1 problem; without a GPU the shards fail with TLPK_NO_DEVICE."""
    sys.path.insert(0, HERE)
    from helpers import block_angular, random_lp_matrix
    A = random_lp_matrix(20, 40, 3, 1)
    with pytest.raises(tk.DimensionMismatch):                       # no row_block
        tk.setup(A, tk.K1(), tk.Backend(device=0, ngpus=2, devices=[0, 0]))
    A, rb = block_angular(nblocks=4, mk=20, nk=40, m0=6, nnz_in=3, link_prob=0.5, seed=2)
    if _lib.lib().tlpk_device_count() == 0:
        with pytest.raises(RuntimeError):
            tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb, ngpus=2, devices=[0, 0]))


def test_julia_glue_matches_the_header_textually():
    """Julia is not in the image, so the glue cannot be executed; what CAN be checked is that the
    `Options` mirror in julia/libtlpk.jl has the fields of `tlpk_options` in the same order with the same
    widths, that every `ccall` names an exported symbol, and that hip.jl reaches the module the way
    Tulip's include order makes it reachable (/root/reference/src/Tulip.jl:18-27)."""
    jl = open(os.path.join(ROOT, "tulip.jl_amd", "julia", "libtlpk.jl")).read()
    hdr = open(os.path.join(ROOT, "include", "tlpk.h")).read()
    body = re.search(r"typedef struct tlpk_options \{(.*?)\} tlpk_options;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    cfields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype = "ptr" if "*" in decl else ("i64" if "int64_t" in decl else "i32")
        for name in decl.replace("*", " ").split()[-1:] if ctype == "ptr" else decl.split(None, 1)[1].split(","):
            cfields.append((name.strip(), ctype))
    jbody = re.search(r"mutable struct Options(.*?)\nend", jl, flags=re.S).group(1)
    jfields = []
    for line in jbody.splitlines():
        mm = re.match(r"\s*(\w+)::(\S+)", line)
        if mm:
            t = mm.group(2)
            jfields.append((mm.group(1), "ptr" if t.startswith("Ptr") else ("i64" if t == "Int64" else "i32")))
    assert jfields == cfields, (jfields, cfields)
    for sym in re.findall(r"ccall\(\(:(\w+),", jl):
        assert hasattr(_lib.lib(), sym), sym
    hip = open(os.path.join(ROOT, "tulip.jl_amd", "julia", "hip.jl")).read()
    assert "using ...TLPLinearAlgebra.LibTLPK" in hip and "using ...LibTLPK" not in hip.replace("using ...TLPLinearAlgebra.LibTLPK", "")
