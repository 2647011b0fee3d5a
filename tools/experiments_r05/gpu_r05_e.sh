#!/bin/bash
# Round 5, GPU session E: k_front_assemble with TALL tiles (4 columns x <= 2304 rows, 73.7 KB of LDS; round 4 measured 16 x 256 tiles slower than
# zero-fill + k_assemble + k_extend_add): parity with the tiles on, A/B on config C4 and the north-star LP.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05e
TLPK_FA_MIN_F=512 TLPK_FA_DENSITY=0 TLPK_POISON=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "large_fronts or block_angular or c4_block_scale or unzeroed or c4_scale_factor or general_sparse or bitwise_deterministic or late_ipm" 2>&1 | tail -3 | tee ${O}_pytest.txt
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernel_ms",{})
print("%.2f ms  runs %s  assemble %s extend_add %s update %s" % (d["ms_per_step"], d["ms_per_step_runs"], k.get("assemble"), k.get("extend_add"), k.get("update")))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi"
for fa in 2048 0 2048 0; do
  echo "c4 TLPK_FA_MIN_F=$fa: $(TLPK_FA_MIN_F=$fa TLPK_FA_DENSITY=0 timeout 300 python bench.py $B 2>/dev/null | python -c "$show")"
done | tee ${O}_fa_tall.txt
for fa in 2048 0; do
  echo "headline TLPK_FA_MIN_F=$fa: $(TLPK_FA_MIN_F=$fa TLPK_FA_DENSITY=0 timeout 400 python bench.py --workload headline --steps 5 --warmup 2 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi 2>/dev/null | python -c "$show")"
done | tee -a ${O}_fa_tall.txt
