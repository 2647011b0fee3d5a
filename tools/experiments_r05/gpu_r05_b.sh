#!/bin/bash
# Round 5, GPU session B: potrf_block_pair (bitwise check + timing in isolation, parity suites, A/B in the library), where the host-pointer
# path's time goes (TLPK_HOSTIO_TIMING), the refinement guard (|r1| must shrink, |r2| must stay) under MPC on the north-star LP,
# the two-shards-on-one-GPU A/B (shard threads x reduction mode), host enqueue time of 8 shards of the north-star shape.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05b
{
for cfg in "1 64" "64 64" "64 256" "64 200" "64 37"; do timeout 60 tools/potrf_wave_bench $cfg 2>&1 | head -1; done
} > ${O}_potrf_pair.txt 2>&1
cat ${O}_potrf_pair.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py -m gpu -x -q 2>&1 | tail -4 > ${O}_pytest.txt; tail -2 ${O}_pytest.txt
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernel_ms",{})
print("%.2f ms  runs %s  potrf %s trsm %s update %s" % (d["ms_per_step"], d["ms_per_step_runs"], k.get("potrf"), k.get("trsm"), k.get("update")))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi"
for pp in 1 0; do
  export TLPK_POTRF_PAIR=$pp
  echo "pds  TLPK_POTRF_PAIR=$pp: $(timeout 300 python bench.py --workload pds --steps 20 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi 2>/dev/null | python -c "$show")"
  echo "c4   TLPK_POTRF_PAIR=$pp: $(timeout 300 python bench.py $B 2>/dev/null | python -c "$show")"
  echo "stair25 TLPK_POTRF_PAIR=$pp: $(timeout 300 python bench.py --workload stair25 --steps 20 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi 2>/dev/null | python -c "$show")"
  echo "rank-local TLPK_POTRF_PAIR=$pp: $(NLIST=8 timeout 200 python tools/rank_local_timing.py 2>&1 | grep nranks)"
done 2>&1 | tee ${O}_potrf_ab.txt
unset TLPK_POTRF_PAIR
TLPK_HOSTIO_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-roofline 2>&1 >/dev/null | grep "host path" | tail -12 | tee ${O}_hostio_c4.txt
TLPK_HOSTIO_TIMING=1 timeout 400 python bench.py --workload headline --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-roofline 2>&1 >/dev/null | grep "host path" | tail -6 | tee ${O}_hostio_headline.txt
ONLY=refine NB=100 timeout 400 python tools/mpc_levers.py 2>&1 | tail -2 | tee ${O}_mpc_refine.txt
for th in 1 0; do for mr in rs gather; do
  echo "two shards on one GPU, TLPK_SHARD_THREADS=$th TLPK_MULTI_REDUCE=$mr: $(TLPK_SHARD_THREADS=$th TLPK_MULTI_REDUCE=$mr NSHARDS=2 timeout 200 python tools/solve_c4_lp.py 2>&1 | grep 'ms per iteration' | head -1)"
done; done 2>&1 | tee ${O}_two_shards.txt
for th in 1 0; do TLPK_SHARD_THREADS=$th NSHARDS=8 timeout 300 python tools/multi_enqueue_timing.py 2>&1 | tail -1; done | tee ${O}_multi_enqueue.txt
