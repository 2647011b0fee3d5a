#!/bin/bash
# Round 5, GPU session G: potrf_block_dpp (TLPK_POTRF_MODE=3) -- microbenchmark against the other 64 x 64 kernels, parity suite with the mode on, A/B in the library.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05g
for a in "1 64" "64 64" "64 256" "64 200" "64 37" "8 600"; do timeout 60 tools/potrf_wave_bench $a; done 2>&1 | tee ${O}_potrf_bench.txt
TLPK_POTRF_MODE=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -x -q 2>&1 | tail -5 | tee ${O}_pytest.txt
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernel_ms",{})
print("%.2f ms  runs %s  potrf %s trsm %s update %s" % (d["ms_per_step"], d["ms_per_step_runs"], k.get("potrf"), k.get("trsm"), k.get("update")))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi"
for m in 3 2; do
  for wl in pds stair25 c4; do
    echo "$wl TLPK_POTRF_MODE=$m: $(TLPK_POTRF_MODE=$m timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "$show")"
  done
  echo "rank-local TLPK_POTRF_MODE=$m: $(TLPK_POTRF_MODE=$m timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1)"
done | tee ${O}_potrf_ab.txt
