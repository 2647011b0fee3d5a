#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06w
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -x -q 2>&1 | tail -3
S="--steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi --no-small-lp --no-c3"
for wl in headline c4; do
timeout 600 python bench.py --workload $wl $S > ${O}_bench_$wl.json 2> ${O}_bench_$wl.err
python - <<P
import json
d=json.load(open("${O}_bench_$wl.json")); sr=d.get("solve_roofline") or {}
print("$wl", round(d["ms_per_step"],3), d.get("ms_per_step_runs"), "solve ms", sr.get("ms_per_solve"), "frac", sr.get("frac"), "pair", (sr.get("pair") or {}).get("ms_over_single"))
print({k: v for k, v in (d.get("kernel_ms") or {}).items()})
P
done
