"""The host side of the drop-in entry points (tlpk_update / tlpk_solve with host vectors): the thread pool and the staging copy of
tulip.jl_amd/csrc/hostcopy.cpp, exercised without a GPU (the HIP side -- bit-identical results of the host-pointer and the
device-pointer calls -- is tests/test_gpu_parity.py::test_host_pointer_calls_equal_device_pointer_calls_bitwise)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.parametrize("threads", ["0", "1", "4", "7"])
def test_host_copy_pool_and_staging_copy(tmp_path, threads):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ in this image")
    exe = str(tmp_path / "hostcopy_check")
    subprocess.check_call([CLANG, "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "hostcopy_check.cpp"),
                           os.path.join(ROOT, "tulip.jl_amd", "csrc", "hostcopy.cpp"), "-o", exe])
    env = dict(os.environ, TLPK_COPY_THREADS=threads)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout
