#!/bin/bash
# A/B of kernel builds: microbenchmark with phase stamps, parity tests, new library against libtlpk_ab_old.so (the previous build of kernels.hip)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06ao
for n in 256 200; do
  echo "== 64 fronts"; timeout 120 ./tools/potrf_wave_bench 64 $n 2>&1 | cut -c1-400
  echo "== one front"; timeout 120 ./tools/potrf_wave_bench 1 $n 2>&1 | cut -c1-400
done > ${O}_potrf_bench.txt 2>&1
grep -v "^64 fronts\|^1 fronts" ${O}_potrf_bench.txt | cut -c1-330
timeout 1200 python -m pytest tests/test_chain.py tests/test_gpu_parity.py tests/test_k2.py tests/test_bump_replay.py -m gpu -x -q 2>&1 | tail -3
S="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
for rep in 1 2; do
for lib in new old; do
  for wl in pds c4; do
    if [ $lib = old ]; then export TLPK_LIB=$PWD/tulip.jl_amd/libtlpk_ab_old.so; else unset TLPK_LIB; fi
    timeout 300 python bench.py --workload $wl $S > ${O}_bench_${wl}_${lib}.json 2> ${O}_bench_${wl}_${lib}.err
    python - <<P
import json
d=json.load(open("${O}_bench_${wl}_${lib}.json")); print("$lib $wl", round(d["ms_per_step"],3), d.get("ms_per_step_runs"))
P
  done
done
done
unset TLPK_LIB
timeout 300 python tools/chain_trace.py > ${O}_chain_trace_pds.txt 2>&1
tail -3 ${O}_chain_trace_pds.txt | cut -c1-250
