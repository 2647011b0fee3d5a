#!/bin/bash
# quick GPU check: core parity tests + C4 bench (no extras)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/quick_bench.json").read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"],2), "kernel_ms", d.get("kernel_ms"), "solve ms", d["solve_roofline"]["ms_per_solve"], "frac", round(d["roofline"]["frac"],4), "frac_step", round(d["roofline"]["frac_step"],4))
P
