#!/bin/bash
# Round 4, GPU session J: extend-add with row bands x narrower column ranges (working set of parent lines per workgroup).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TLPK_EA_BANDS=3 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c4_scale or large_fronts" 2>&1 | tail -2 | cut -c1-200
B="--steps 5 --warmup 2 --no-cpu-baseline --no-host-abi --no-small-lp --no-headline --no-c3"
run() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --workload headline $B > gpurun_out/r04j_h.json 2> gpurun_out/r04j_h.err
  env "$@" timeout 300 python bench.py $B > gpurun_out/r04j_c.json 2> gpurun_out/r04j_c.err
  python - "$label" <<'P'
import json, sys
out = [sys.argv[1]]
for f, nm in (("gpurun_out/r04j_c.json", "c4"), ("gpurun_out/r04j_h.json", "headline")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernel_ms"]
        out.append(f"{nm}: ms/step {d['ms_per_step']:.2f} ea {k['extend_add']:.2f} update {k['update']:.2f}")
    except Exception as e:
        out.append(f"{nm}: FAILED {e}")
print(" | ".join(out))
P
}
run "default (16 columns, 1 band)" A=1
run "bands=2" TLPK_EA_BANDS=2
run "bands=4" TLPK_EA_BANDS=4
run "cols=4" TLPK_EA_COLS=4
run "cols=4 bands=2" TLPK_EA_COLS=4 TLPK_EA_BANDS=2
run "cols=4 bands=4" TLPK_EA_COLS=4 TLPK_EA_BANDS=4
run "cols=8 bands=2" TLPK_EA_COLS=8 TLPK_EA_BANDS=2
