#!/bin/bash
# Round 3, GPU session E: whole GPU suite on the tail-split schedule + bench with / without the tail split.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/r03_e_pytest.txt 2>&1
tail -12 gpurun_out/r03_e_pytest.txt
B="--no-cpu-baseline --no-small-lp --no-host-abi"
for s in 0 512; do
  TLPK_TAIL_SLOTS=$s timeout 600 python bench.py --steps 20 --warmup 5 $B > gpurun_out/r03_e_bench_tail$s.json 2> gpurun_out/r03_e_bench_tail$s.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03_e_bench_tail$s.json"))
print("TAIL_SLOTS=$s: ms/step %.2f unpaired %.2f frac %.3f frac_exec %.3f kernel_ms %s | headline %.2f frac %.3f" % (d["ms_per_step"], d.get("unpaired_ms_per_step", 0), d["roofline"]["frac"], d["roofline"]["frac_executed"], {k: d["kernel_ms"][k] for k in ("update", "update_reduce", "extend_add", "trsm", "potrf")}, d["headline"]["ms_per_step"], d["headline"]["roofline"]["frac"]))
PY
done
