#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06aa
S="--steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi --no-small-lp --no-c3"
for rep in 1 2; do
for v in occ2 occ3; do
cp tools/ab/libtlpk_$v.so tulip.jl_amd/libtlpk.so
for wl in c4 headline; do
timeout 600 python bench.py --workload $wl $S > ${O}_bench_${wl}_$v.json 2> ${O}_bench_${wl}_$v.err
python - <<P
import json
d=json.load(open("${O}_bench_${wl}_$v.json")); sr=d.get("solve_roofline") or {}
print("$v $wl", round(d["ms_per_step"],3), "solve ms", round(sr.get("ms_per_solve"),4), "pair ms", round((sr.get("pair") or {}).get("ms"),4), "pair/single", round((sr.get("pair") or {}).get("ms_over_single"),4))
P
done
done
done
