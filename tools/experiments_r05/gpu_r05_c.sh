#!/bin/bash
# Round 5, GPU session C: potrf_block_pair bit for bit against potrf_block in isolation; host-pointer path with 2 MB DMA groups / 512 KB staging
# chunks (timing breakdown, bitwise test, host_abi); graph replay of the single-stream solve schedule on a block-angular handle (A/B).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05c
for cfg in "1 64" "64 64" "64 256" "64 200" "64 37"; do timeout 60 tools/potrf_wave_bench $cfg 2>&1 | head -1; done | tee ${O}_potrf_pair.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_pointer or refinement or graph_replay or two_right_hand or copies_inputs or golden or not_posdef" 2>&1 | tail -3 | tee ${O}_pytest.txt
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("%.2f ms  runs %s  host_abi %s  unpaired %s" % (d["ms_per_step"], d["ms_per_step_runs"], d.get("host_abi",{}).get("ms_per_step"), d.get("unpaired_ms_per_step")))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-roofline"
for g in 1 0; do echo "c4 TLPK_GRAPH_SOLVE=$g: $(TLPK_GRAPH_SOLVE=$g timeout 300 python bench.py $B 2>/dev/null | python -c "$show")"; done | tee ${O}_graph_solve.txt
for g in 1 0; do echo "headline TLPK_GRAPH_SOLVE=$g: $(TLPK_GRAPH_SOLVE=$g timeout 400 python bench.py --workload headline --steps 5 --warmup 2 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-roofline 2>/dev/null | python -c "$show")"; done | tee -a ${O}_graph_solve.txt
for g in 1 0; do TLPK_GRAPH_SOLVE=$g TLPK_HOSTIO_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-roofline 2>&1 >/dev/null | grep "host path" | tail -6; done | tee ${O}_hostio_c4.txt
