#!/bin/bash
# GPU call D: backward-sweep chain rework + device-resident HSD iterate
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sweeps or deterministic or golden or random_sparse" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_hsd_device.py -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r2d_hsd.log; tail -5 gpurun_out/r2d_hsd.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r2d_tests.log; tail -3 gpurun_out/r2d_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err
timeout 600 python bench.py --workload c3 --c3-rows 50000 --steps 3 --warmup 1 --no-host-abi > gpurun_out/r2d_bench_c3.json 2> gpurun_out/r2d_bench_c3.err
python - <<'P'
import json
for f in ("r2d_bench","r2d_bench_c3"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step", round(d["ms_per_step"],2), "kernel_ms", d.get("kernel_ms"), "solve", d.get("solve_roofline",{}))
    except Exception as e: print(f, "ERR", e)
P
