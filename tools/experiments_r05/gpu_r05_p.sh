#!/bin/bash
# Round 5, GPU session P: the left-looking products in front of a diagonal block with two operand register sets (the next k-step requested before the
# products of the current one); the kernel-agreement test (modes 3 / 2 / 0 on one handle, K1 and K2); pds / C4 / C3 in the library.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05p
for a in "1 64" "64 256" "64 200"; do timeout 60 tools/potrf_wave_bench $a | head -1 | sed 's/; max |L0 - L_wave.*NaNs/; NaNs/'; done 2>&1 | tee ${O}_potrf_bench.txt
for a in "1 256"; do timeout 60 tools/potrf_wave_bench_trace $a | tail -1; done 2>&1 | tee -a ${O}_potrf_bench.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "diagonal_block_kernels_agree" 2>&1 | grep "max |L3\|passed\|failed\|Error\|assert" | cut -c1-300 | tee ${O}_agree.txt
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernel_ms",{})
print("%.2f ms  runs %s  potrf %s trsm %s update %s" % (d["ms_per_step"], d["ms_per_step_runs"], k.get("potrf"), k.get("trsm"), k.get("update")))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi"
for m in 3 2 3; do
  for wl in pds c4; do
    echo "$wl TLPK_POTRF_MODE=$m: $(TLPK_POTRF_MODE=$m timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "$show")"
  done
done | tee ${O}_potrf_ab.txt
echo "rank-local: $(timeout 300 python tools/rank_local_timing.py 2>&1 | tail -1)" | tee -a ${O}_potrf_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_hsd_device.py -m gpu -q 2>&1 | tail -4 | tee ${O}_pytest.txt
