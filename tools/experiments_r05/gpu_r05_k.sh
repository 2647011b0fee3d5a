#!/bin/bash
# Round 5, GPU session K: potrf_block_dpp with the ROW-oriented diagonal block (one v_fmac_f64_dpp per entry, multipliers from one triangle only):
# microbenchmark + phase stamps, device-loop tests (K2 iteration counts), parity, A/B in the library (3 = one trsm call per 64 rows, 4 = per step).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05k
for a in "1 64" "64 64" "64 256" "64 200" "64 37"; do timeout 60 tools/potrf_wave_bench $a | head -1; done 2>&1 | tee ${O}_potrf_bench.txt
for a in "1 64" "1 256"; do timeout 60 tools/potrf_wave_bench_trace $a | tail -1; done 2>&1 | tee -a ${O}_potrf_bench.txt
timeout 600 python -m pytest tests/test_hsd_device.py -m gpu -q -k "multi_device_handle" -s 2>&1 | grep "one device\|assert \|Error\|passed\|failed" | cut -c1-250 | tee ${O}_device_loops.txt
TLPK_POTRF_MODE=4 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_hsd_device.py tests/test_mpc_device.py -m gpu -q 2>&1 | tail -5 | tee ${O}_pytest.txt
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernel_ms",{})
print("%.2f ms  runs %s  potrf %s trsm %s update %s" % (d["ms_per_step"], d["ms_per_step_runs"], k.get("potrf"), k.get("trsm"), k.get("update")))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi"
for m in 3 4 2 4; do
  for wl in pds stair25 c4; do
    echo "$wl TLPK_POTRF_MODE=$m: $(TLPK_POTRF_MODE=$m timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "$show")"
  done
done | tee ${O}_potrf_ab.txt
for m in 4 2; do echo "rank-local TLPK_POTRF_MODE=$m: $(TLPK_POTRF_MODE=$m timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1)"; done | tee -a ${O}_potrf_ab.txt
