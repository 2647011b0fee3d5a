"""bench.py prints ONE JSON line with the agreed keys (the driver parses it); the cpu_baseline leg runs the comparator in a
child process on a bounded sample.  Small step counts: this checks the contract, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_contract():
    d = run_bench("--steps", "2", "--warmup", "1", "--no-headline", "--no-cpu-baseline", "--no-small-lp")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "solve_roofline", "host_abi", "kernel_ms"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # the timed region is run `--repeats` (3) times: ms_per_step is the median run, all runs are in the line
    runs = d["ms_per_step_runs"]
    assert len(runs) == 3 and min(runs) == d["ms_per_step_min"] and max(runs) == d["ms_per_step_max"] and sorted(runs)[1] == pytest.approx(d["ms_per_step"], rel=1e-3)
    assert d["host_abi"]["copy_threads"] >= 1 and d["host_abi"]["steps"] >= 2
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-9 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12
    assert 0.2 < r["frac"] < 1.0 and r["traffic"] is None or r["traffic"] > 0
    assert max(d["config"]["residual_inf"]) <= 1e-6                        # the timed step solved the system
    s = d["solve_roofline"]
    assert s["bound"] == "hbm" and s["unit"] == "GB/s" and 0.05 < s["frac"] < 1.0


@pytest.mark.gpu
def test_bench_cpu_baseline_leg():
    d = run_bench("--steps", "2", "--warmup", "1", "--no-headline", "--no-host-abi", "--no-roofline", "--no-small-lp")
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, (k, c)
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"] and c["value"] > 0
    assert max(c["residual_inf"]) <= 1e-6                                  # the comparator solved the same system
    # BASELINE.md B2 (SciPy splu on one diagonal block: fill within a few per cent of this library's AMD ordering) and the B4 probe
    t = c["third_party"]
    assert "error" not in t, t
    assert 0.7 < t["fill_ratio_superlu_over_this"] < 1.4 and t["seconds_factor"] > 0 and t["residual_inf"] <= 1e-6
    assert isinstance(c["reference_cpu"], (str, dict)) and (isinstance(c["reference_cpu"], dict) or "julia" in c["reference_cpu"])


@pytest.mark.gpu
def test_bench_small_lp_legs():
    """The latency-bound configs (25fv47 class, pds-20 class): GPU step, host-ABI step and the CPU comparator, with the statement."""
    d = run_bench("--steps", "2", "--warmup", "1", "--no-headline", "--no-roofline", "--no-host-abi", "--blocks", "4")
    for wl in ("stair25", "pds"):
        leg = d["small_lp"][wl]
        assert "error" not in leg, leg
        assert leg["ms_per_step"] > 0 and leg["cpu_ms_per_step"] > 0 and leg["statement"].startswith("latency-bound")
        assert max(leg["residual_inf"]) <= 1e-6


@pytest.mark.gpu
def test_bench_multi_gpu_branch_with_a_world_of_one():
    """The branch the 8-GPU scaling run takes -- process group on backend nccl (= RCCL), split-phase update / solve, all-reduce of
    the root panel and the root right-hand side on torch's stream ordered with the library stream by events -- executed with one
    rank on this box's single GPU: same step, same residuals as the plain N = 1 path."""
    a = run_bench("--steps", "2", "--warmup", "1", "--no-headline", "--no-cpu-baseline", "--no-small-lp", "--no-roofline", "--no-host-abi", "--blocks", "8")
    b = run_bench("--steps", "2", "--warmup", "1", "--blocks", "8", "--force-collectives")
    assert "collectives" in b and b["n_gpus"] == 1
    assert max(b["config"]["residual_inf"]) <= 1e-6 and max(a["config"]["residual_inf"]) <= 1e-6
    assert b["ms_per_step"] < 3 * a["ms_per_step"] + 5.0                     # no host round trips hidden in the collective path


@pytest.mark.gpu
def test_two_ranks_on_one_device_are_refused_cleanly():
    """The first multi-GPU launch must fail loudly, not hang: `torchrun --nproc-per-node 2 bench.py --gpus 2 --blocks 8` with BOTH ranks on device 0 (this box
    has one GPU; RCCL refuses duplicate devices when the communicator is created).  bench.py builds the communicator right behind init_process_group and turns
    the refusal into one clear line on stderr and exit code 3 -- within the timeout, no JSON line on stdout."""
    import socket
    import subprocess
    import time
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, TLPK_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", TORCH_NCCL_ASYNC_ERROR_HANDLING="1", NCCL_DEBUG="WARN")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--blocks", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    t0 = time.time()
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    took = time.time() - t0
    err = p.stderr.decode(errors="replace")
    assert p.returncode != 0, "two ranks on one device must not produce a bench line"
    assert "RCCL refused the communicator" in err or "Duplicate GPU" in err or "duplicate" in err.lower(), err[-3000:]
    assert not any(line.startswith("{") for line in p.stdout.decode(errors="replace").splitlines()), "no JSON line from a refused launch"
    assert took < 240
