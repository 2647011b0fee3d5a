#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06al
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_hsd_device.py tests/test_lp_configs.py -m gpu -x -q 2>&1 | tail -3
S="--steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi --no-small-lp --no-c3"
run() { local name=$1; local wl=$2; shift; shift
    env "$@" timeout 600 python bench.py --workload $wl $S > ${O}_bench_${wl}_$name.json 2> ${O}_bench_${wl}_$name.err
    python - <<P
import json
d=json.load(open("${O}_bench_${wl}_$name.json")); sr=d.get("solve_roofline") or {}; km=d.get("kernel_ms") or {}
print("$wl $name", round(d["ms_per_step"],3), "solve ms", round(sr.get("ms_per_solve"),4), "frac", round(sr.get("frac"),4), "pair", round((sr.get("pair") or {}).get("ms"),4), "launches", d["config"].get("launches_solve"))
P
}
for rep in 1 2; do
run gat headline X=1
run nogat headline TLPK_SMALL_GATHER=0
run gat c4 X=1
run nogat c4 TLPK_SMALL_GATHER=0
run gat stair25 X=1
run nogat stair25 TLPK_SMALL_GATHER=0
run gat pds X=1
run nogat pds TLPK_SMALL_GATHER=0
done
