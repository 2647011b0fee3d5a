#!/bin/bash
# Round 4, GPU session D: front assembly of the root front only (dense contributions), A/B.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c4_scale or block_angular or sharded" 2>&1 | tail -3 | cut -c1-300
B="--steps 10 --warmup 3 --unpaired --no-cpu-baseline --no-host-abi --no-small-lp --no-headline"
for fa in 512 0; do
  TLPK_FA_MIN_F=$fa timeout 300 python bench.py $B > gpurun_out/r04d_c4_fa$fa.json 2> gpurun_out/r04d_c4_fa$fa.err
  TLPK_FA_MIN_F=$fa timeout 300 python bench.py --workload headline --steps 5 --warmup 2 --unpaired --no-cpu-baseline --no-host-abi --no-small-lp > gpurun_out/r04d_headline_fa$fa.json 2> gpurun_out/r04d_headline_fa$fa.err
done
python - <<'P'
import json
for f in ("r04d_c4_fa512", "r04d_c4_fa0", "r04d_headline_fa512", "r04d_headline_fa0"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e, open(f"gpurun_out/{f}.err").read()[-600:]); continue
    r = d["roofline"]
    print(f, "ms/step", round(d["ms_per_step"], 2), "frac", round(r["frac"], 4), "frac_step", round(r["frac_step"], 4), "kernel_ms", {k: round(v, 2) for k, v in d.get("kernel_ms", {}).items()})
P
