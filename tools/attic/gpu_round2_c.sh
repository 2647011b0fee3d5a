#!/bin/bash
# GPU call C: where does the sweep time go?  polling back-off sweep + PMC on the sweep kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-headline --no-host-abi"
for P in "8,16,32" "4,16,64" "8,8,128" "2,64,16" "16,1000000,16" "2,1000000,2"; do
  TLPK_POLL=$P timeout 300 $B > gpurun_out/r2c_poll_$P.json 2>/dev/null
  python - "$P" <<'PY'
import json,sys
P=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/r2c_poll_%s.json"%P).read().strip().splitlines()[-1])
    print("poll",P,"ms/step",round(d["ms_per_step"],2),"fwd",d["kernel_ms"]["solve_fwd"],"bwd",d["kernel_ms"]["solve_bwd"])
except Exception as e: print("poll",P,"ERR",e)
PY
done
TLPK_POLL=8,16,32 timeout 300 python bench.py --workload c3 --c3-rows 50000 --steps 2 --warmup 1 --no-host-abi > gpurun_out/r2c_c3.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2c_c3.json").read().strip().splitlines()[-1]); print("c3", d["ms_per_step"], d["kernel_ms"])
PY
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/r2c_pmc1 -- \
    python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi > gpurun_out/r2c_pmc1.log 2>&1
TLPK_STREAMS=1 TLPK_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/r2c_pmc2 -- \
    python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-headline --no-host-abi > gpurun_out/r2c_pmc2.log 2>&1
python tools/pmc_summarise.py gpurun_out/r2c_pmc1 gpurun_out/r2c_pmc2 2>&1 | grep -E "sweep|kernel \||---" 
