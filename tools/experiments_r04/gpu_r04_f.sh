#!/bin/bash
# Round 4, GPU session F: small-front solve kernels (templated bodies), C3 test + bench leg, MPC levers (K2 / refinement) on the north-star family.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c3_shape or small or singleton or golden or block_angular or two_right" 2>&1 | tail -4 | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-abi --no-small-lp > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r04f_bench.json").read().strip().splitlines()[-1])
print("c4 ms/step", round(d["ms_per_step"], 2), "unpaired", round(d["unpaired_ms_per_step"], 2), "kernel_ms", {k: round(v, 2) for k, v in d["kernel_ms"].items()}, "solve", round(d["solve_roofline"]["ms_per_solve"], 3))
h = d["headline"]; print("headline ms/step", round(h["ms_per_step"], 2), "kernel_ms", {k: round(v, 2) for k, v in h["kernel_ms"].items()}, "solve", round(h["solve_roofline"]["ms_per_solve"], 3), "frac", round(h["roofline"]["frac"], 4))
c = d.get("c3", {}); print("c3", {k: c.get(k) for k in ("ms_per_step", "frac_step", "error")}, "frac", c.get("roofline", {}).get("frac"), "solve", c.get("solve_roofline", {}).get("ms_per_solve"))
P
NB=8 timeout 900 python tools/mpc_levers.py > gpurun_out/r04f_mpc_levers_8blocks.txt 2>&1; cat gpurun_out/r04f_mpc_levers_8blocks.txt | cut -c1-330
