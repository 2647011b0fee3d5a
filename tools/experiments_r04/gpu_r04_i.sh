#!/bin/bash
# Round 4, GPU session I: amalgamation constants again, now that k_update skips structurally zero slabs (padding is cheaper than in round 3).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 5 --warmup 2 --no-cpu-baseline --no-host-abi --no-small-lp --no-headline --no-c3"
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --workload headline $B > gpurun_out/r04i_h.json 2> gpurun_out/r04i_h.err
  env "$@" timeout 300 python bench.py $B > gpurun_out/r04i_c.json 2> gpurun_out/r04i_c.err
  python - "$label" <<'P'
import json, sys
out = [sys.argv[1]]
for f, nm in (("gpurun_out/r04i_c.json", "c4"), ("gpurun_out/r04i_h.json", "headline")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]; k = d["kernel_ms"]
        out.append(f"{nm}: ms/step {d['ms_per_step']:.2f} frac {r['frac']:.4f} exec/alg {r['flops_executed_per_step'] / r['flops_per_step']:.3f} update {k['update']:.2f} ea {k['extend_add']:.2f} trsm {k['trsm']:.2f} stored/nnzL {d['config']['stored_over_nnzL']:.3f}")
    except Exception as e:
        out.append(f"{nm}: FAILED {e}")
print(" | ".join(out))
P
}
run "default (GAMMA_TALL=400 TALL_RATIO=0.5 GAMMA=25)" A=1
run "GAMMA_TALL=800" TLPK_RELAX_GAMMA_TALL=800
run "GAMMA_TALL=200" TLPK_RELAX_GAMMA_TALL=200
run "TALL_RATIO=0.4" TLPK_RELAX_TALL_RATIO=0.4
run "TALL_RATIO=0.6" TLPK_RELAX_TALL_RATIO=0.6
run "TALL_RATIO=0.4 GAMMA_TALL=800" TLPK_RELAX_TALL_RATIO=0.4 TLPK_RELAX_GAMMA_TALL=800
run "ZFRAC=0 (no unbounded-width zero-fraction class)" TLPK_RELAX_ZFRAC=0
run "GAMMA=50" TLPK_RELAX_GAMMA=50
