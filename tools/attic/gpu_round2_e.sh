#!/bin/bash
# GPU call E: K2 (augmented system, signed Cholesky) on the device + regression of everything else
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_k2.py -m gpu -q -x -s 2>&1 | tail -30 > gpurun_out/r2e_k2.log; tail -8 gpurun_out/r2e_k2.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r2e_tests.log; tail -3 gpurun_out/r2e_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi > gpurun_out/r2e_bench.json 2> gpurun_out/r2e_bench.err
python - <<'P'
import json
for f in ("r2e_bench",):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step", round(d["ms_per_step"],2), "kernel_ms", d.get("kernel_ms"))
    except Exception as e: print(f, "ERR", e)
P
