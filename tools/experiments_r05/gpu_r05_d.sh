#!/bin/bash
# Round 5, GPU session D: host-pointer path with the DMA groups alternating between two copy streams (A/B against one stream), bitwise test.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05d
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -m gpu -x -q -k "host_pointer or copies_inputs or golden or plain_c or conformance" 2>&1 | tail -3 | tee ${O}_pytest.txt
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("%.2f ms  runs %s  host_abi %s  unpaired %s" % (d["ms_per_step"], d["ms_per_step_runs"], d.get("host_abi",{}).get("ms_per_step"), d.get("unpaired_ms_per_step")))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-roofline"
for cs in 2 1 2 1; do echo "c4 TLPK_COPY_STREAMS=$cs: $(TLPK_COPY_STREAMS=$cs timeout 300 python bench.py $B 2>/dev/null | python -c "$show")"; done | tee ${O}_copy_streams.txt
for cs in 2 1; do echo "headline TLPK_COPY_STREAMS=$cs: $(TLPK_COPY_STREAMS=$cs timeout 400 python bench.py --workload headline --steps 5 --warmup 2 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-roofline 2>/dev/null | python -c "$show")"; done | tee -a ${O}_copy_streams.txt
for cs in 2 1; do TLPK_COPY_STREAMS=$cs TLPK_HOSTIO_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-roofline 2>&1 >/dev/null | grep "host path" | tail -5; done | tee ${O}_hostio_c4.txt
