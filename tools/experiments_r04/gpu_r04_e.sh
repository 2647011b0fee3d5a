#!/bin/bash
# Round 4, GPU session E: barrier-free 64 x 64 diagonal-block factorisation (potrf_block_wave) -- parity, A/B on the chain-bound workloads.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
for wv in 1 0; do
  echo "== TLPK_POTRF_WAVE=$wv"
  TLPK_POTRF_WAVE=$wv timeout 300 python tools/small_lp_timing.py 2>&1 | grep -v "^$" | tail -6 | cut -c1-400
  TLPK_POTRF_WAVE=$wv NLIST=1,8 timeout 300 python tools/rank_local_timing.py 2>&1 | tail -4 | cut -c1-400
  TLPK_POTRF_WAVE=$wv timeout 300 python bench.py --steps 10 --warmup 3 --unpaired --no-cpu-baseline --no-host-abi --no-small-lp --no-headline > gpurun_out/r04e_c4_wave$wv.json 2> gpurun_out/r04e_c4_wave$wv.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r04e_c4_wave$wv.json").read().strip().splitlines()[-1])
print("c4 ms/step", round(d["ms_per_step"], 2), "kernel_ms", {k: round(v, 2) for k, v in d.get("kernel_ms", {}).items()})
P
done
