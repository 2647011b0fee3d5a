#!/bin/bash
# Round 4, GPU session L: A/B of two builds of the library (tools/ab/*.so, TLPK_LIB) -- k_trsm variants.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 8 --warmup 2 --no-cpu-baseline --no-host-abi --no-small-lp --no-headline --no-c3"
for rep in 1 2; do
for lib in ${LIBS:-base trsm1}; do
  export TLPK_LIB=$GRAFT_REPO_ROOT/tools/ab/libtlpk_$lib.so
  timeout 300 python bench.py --workload headline $B > gpurun_out/r04l_h.json 2> gpurun_out/r04l_h.err
  timeout 300 python bench.py $B > gpurun_out/r04l_c.json 2> gpurun_out/r04l_c.err
  python - "$lib" <<'P'
import json, sys
out = [sys.argv[1]]
for f, nm in (("gpurun_out/r04l_c.json", "c4"), ("gpurun_out/r04l_h.json", "headline")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernel_ms"]
        out.append(f"{nm}: ms/step {d['ms_per_step']:.2f} " + " ".join(f"{a} {b:.2f}" for a, b in k.items() if isinstance(b, (int, float))))
    except Exception as e:
        out.append(f"{nm}: FAILED {e!r}"); print(open(f.replace('.json', '.err')).read()[-800:])
print(" | ".join(out))
P
done
done
unset TLPK_LIB
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
