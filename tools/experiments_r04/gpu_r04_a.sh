#!/bin/bash
# Round 4, GPU session A: skip lists of structural zeros -- parity subset, default bench, A/B on the headline instance.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -q -x -k "skip_lists or golden or large_fronts or c4_scale or deterministic or random_sparse or macro" 2>&1 | tail -6 | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err; tail -3 gpurun_out/r04a_bench.err | cut -c1-300
for skip in 0 1; do
  TLPK_SKIP=$skip timeout 300 python bench.py --workload headline --steps 5 --warmup 2 --unpaired --no-cpu-baseline --no-host-abi --no-small-lp > gpurun_out/r04a_headline_skip$skip.json 2> gpurun_out/r04a_headline_skip$skip.err
done
TLPK_SKIP=0 timeout 300 python bench.py --steps 10 --warmup 3 --unpaired --no-cpu-baseline --no-host-abi --no-small-lp --no-headline > gpurun_out/r04a_c4_skip0.json 2> gpurun_out/r04a_c4_skip0.err
python - <<'P'
import json
for f in ("r04a_bench", "r04a_headline_skip0", "r04a_headline_skip1", "r04a_c4_skip0"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r = d["roofline"]
    print(f, "ms/step", round(d["ms_per_step"], 2), "unpaired", d.get("unpaired_ms_per_step"), "frac", round(r["frac"], 4), "frac_exec", round(r.get("frac_executed", 0), 4),
          "frac_step", round(r["frac_step"], 4), "kernel_ms", {k: round(v, 2) for k, v in d.get("kernel_ms", {}).items()} if isinstance(d.get("kernel_ms"), dict) else d.get("kernel_ms"))
    if "headline" in d and isinstance(d["headline"], dict):
        h = d["headline"]; print("   headline:", {k: h[k] for k in h if k in ("ms_per_step", "roofline_frac", "frac_executed", "frac_step", "gpu_over_cpu")})
P
