#!/bin/bash
# Round 4, GPU session C: front assembly (k_front_assemble) -- parity (also on NaN-poisoned storage), A/B against zero-fill + k_assemble + k_extend_add.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
K="skip_lists or golden or large_fronts or c4_scale or deterministic or random_sparse or macro or block_angular or sharded or multi_device"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -q -x -k "$K" 2>&1 | tail -6 | cut -c1-300
TLPK_POISON=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -q -x -k "golden or large_fronts or c4_scale or block_angular" 2>&1 | tail -4 | cut -c1-300
B="--steps 10 --warmup 3 --unpaired --no-cpu-baseline --no-host-abi --no-small-lp --no-headline"
for fa in 512 0 2048; do
  TLPK_FA_MIN_F=$fa timeout 300 python bench.py $B > gpurun_out/r04c_c4_fa$fa.json 2> gpurun_out/r04c_c4_fa$fa.err
  TLPK_FA_MIN_F=$fa timeout 300 python bench.py --workload headline --steps 5 --warmup 2 --unpaired --no-cpu-baseline --no-host-abi --no-small-lp > gpurun_out/r04c_headline_fa$fa.json 2> gpurun_out/r04c_headline_fa$fa.err
done
python - <<'P'
import json
for f in ("r04c_c4_fa512", "r04c_c4_fa0", "r04c_c4_fa2048", "r04c_headline_fa512", "r04c_headline_fa0", "r04c_headline_fa2048"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e, open(f"gpurun_out/{f}.err").read()[-600:]); continue
    r = d["roofline"]
    print(f, "ms/step", round(d["ms_per_step"], 2), "frac", round(r["frac"], 4), "frac_step", round(r["frac_step"], 4), "kernel_ms", {k: round(v, 2) for k, v in d.get("kernel_ms", {}).items()})
P
