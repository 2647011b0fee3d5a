#!/bin/bash
# k_update task order / XCD mapping experiment: time + FETCH_SIZE per setting
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TLPK_UPD_REMAP=2 TLPK_UPD_SUPER=8 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c4_block_scale or large_fronts or c3_shape" 2>&1 | tail -2
B="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi"
for cfg in "0 1" "2 1" "2 8" "1 1" "2 4"; do
  set -- $cfg
  export TLPK_UPD_REMAP=$1 TLPK_UPD_SUPER=$2
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi > gpurun_out/remap_$1_$2.json 2> gpurun_out/remap_$1_$2.err
  python - "$1" "$2" <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/remap_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print("remap",sys.argv[1],"super",sys.argv[2],"ms/step", round(d["ms_per_step"],2), "update", d["kernel_ms"]["update"], "frac", round(d["roofline"]["frac"],4))
P
done
for cfg in "0 1" "2 8"; do
  set -- $cfg
  export TLPK_UPD_REMAP=$1 TLPK_UPD_SUPER=$2
  TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/remap_pmc_$1_$2 -- $B > gpurun_out/remap_pmc_$1_$2.log 2>&1
  echo "PMC remap $1 super $2"; python tools/pmc_summarise.py gpurun_out/remap_pmc_$1_$2 | grep -E "k_update \|"
done
