#!/bin/bash
# Round 5, GPU session I: potrf_block_dpp with bit-identical triangles of the diagonal block (product-first update) + one trsm call per step inside
# k_potrf_wide: microbenchmark, the device-loop tests that showed the K2 accuracy loss of the multiplier form, parity, A/B in the library.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05i
for a in "1 64" "64 64" "64 256" "64 200" "64 37"; do timeout 60 tools/potrf_wave_bench $a | head -1; done 2>&1 | tee ${O}_potrf_bench.txt
for a in "1 64" "1 256"; do timeout 60 tools/potrf_wave_bench_trace $a | tail -1; done 2>&1 | tee -a ${O}_potrf_bench.txt
timeout 600 python -m pytest tests/test_hsd_device.py -m gpu -q -k "multi_device_handle" -s 2>&1 | grep "one device\|assert \|Error\|passed\|failed" | cut -c1-400 | tee ${O}_device_loops.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_lp_configs.py -m gpu -q 2>&1 | tail -5 | tee ${O}_pytest.txt
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernel_ms",{})
print("%.2f ms  runs %s  potrf %s trsm %s update %s" % (d["ms_per_step"], d["ms_per_step_runs"], k.get("potrf"), k.get("trsm"), k.get("update")))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi"
for wl in pds stair25 c4; do
  echo "$wl: $(timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "$show")"
done | tee ${O}_potrf_ab.txt
echo "rank-local: $(timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1)" | tee -a ${O}_potrf_ab.txt
