#!/bin/bash
# GPU call B of round 2: persistent solve sweeps -- parity suite, A/B bench (sweeps on / off)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "sweeps or deterministic or golden or random_sparse" 2>&1 | tail -40 > gpurun_out/r2b_tests_quick.log
tail -3 gpurun_out/r2b_tests_quick.log
if grep -q "failed\|error\|Timeout\|Aborted" gpurun_out/r2b_tests_quick.log; then echo "quick tests failed: stopping"; exit 1; fi
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -60 > gpurun_out/r2b_tests.log
tail -3 gpurun_out/r2b_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
TLPK_SWEEP=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi > gpurun_out/r2b_bench_nosweep.json 2> gpurun_out/r2b_bench_nosweep.err
timeout 600 python bench.py --workload c3 --c3-rows 50000 --steps 3 --warmup 1 --no-host-abi > gpurun_out/r2b_bench_c3.json 2> gpurun_out/r2b_bench_c3.err
python - <<'P'
import json
for f in ("r2b_bench","r2b_bench_nosweep","r2b_bench_c3"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step", round(d["ms_per_step"],2), "kernel_ms", d.get("kernel_ms"), "solve", d.get("solve_roofline",{}).get("ms_per_solve"), "headline", (d.get("headline") or {}).get("ms_per_step"), (d.get("headline") or {}).get("solve_roofline",{}).get("ms_per_solve"))
    except Exception as e: print(f, "ERR", e)
P
