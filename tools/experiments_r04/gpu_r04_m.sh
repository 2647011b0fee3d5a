#!/bin/bash
# Round 4, GPU session M: knob sweep on the final code -- stream groups, super-tile size of the update tile order.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 8 --warmup 2 --no-cpu-baseline --no-host-abi --no-small-lp --no-headline --no-c3 --no-roofline"
run() {
  timeout 300 python bench.py --workload headline $B > gpurun_out/r04m_h.json 2> gpurun_out/r04m_h.err
  timeout 300 python bench.py $B > gpurun_out/r04m_c.json 2> gpurun_out/r04m_c.err
  python - "$1" <<'P'
import json, sys
out = [sys.argv[1]]
for f, nm in (("gpurun_out/r04m_c.json", "c4"), ("gpurun_out/r04m_h.json", "headline")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        out.append(f"{nm}: ms/step {d['ms_per_step']:.2f}")
    except Exception as e:
        out.append(f"{nm}: FAILED {e!r}")
print(" | ".join(out))
P
}
run "default"
for s in 1 3 4; do TLPK_STREAMS=$s run "TLPK_STREAMS=$s"; done
for u in 2 8; do TLPK_UPD_SUPER=$u run "TLPK_UPD_SUPER=$u"; done
TLPK_SKIP_MIN_F=128 run "TLPK_SKIP_MIN_F=128"
TLPK_MACRO_TILES=4096 run "TLPK_MACRO_TILES=4096"
run "default"
