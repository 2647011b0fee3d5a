#!/bin/bash
# Round 4, GPU session K: longest-first order of the k_update tiles of a launch.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 8 --warmup 2 --no-cpu-baseline --no-host-abi --no-small-lp --no-headline --no-c3"
for lpt in 0 1 0 1; do
  TLPK_UPD_LPT=$lpt timeout 300 python bench.py --workload headline $B > gpurun_out/r04k_h.json 2> gpurun_out/r04k_h.err
  TLPK_UPD_LPT=$lpt timeout 300 python bench.py $B > gpurun_out/r04k_c.json 2> gpurun_out/r04k_c.err
  python - "lpt=$lpt" <<'P'
import json, sys
out = [sys.argv[1]]
for f, nm in (("gpurun_out/r04k_c.json", "c4"), ("gpurun_out/r04k_h.json", "headline")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernel_ms"]
    out.append(f"{nm}: ms/step {d['ms_per_step']:.2f} update {k['update']:.2f} frac {d['roofline']['frac']:.4f}")
print(" | ".join(out))
P
done
