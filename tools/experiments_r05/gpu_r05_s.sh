#!/bin/bash
# Round 5, GPU session S: does the diagonal-block chain of the pds-class LP run beside the rows-below update?  Kernel trace of the default schedule (graph replay off so that
# every launch is a trace record) + the sweep-line tool; the same with the side stream at high priority (TLPK_SIDE_PRIO=1).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05s
S="--workload pds --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --no-c3"
for prio in 0 1; do
  TLPK_SIDE_PRIO=$prio TLPK_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --output-format csv -d ${O}_trace$prio -- python bench.py $S > ${O}_trace$prio.log 2>&1
  STEP=4 python tools/timeline_overlap.py $(ls ${O}_trace$prio/*/*kernel_trace.csv | head -1) > ${O}_timeline_pds_prio$prio.txt 2>&1
  tail -14 ${O}_timeline_pds_prio$prio.txt
  rm -rf ${O}_trace$prio
done
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.2f ms  runs %s" % (d["ms_per_step"], d["ms_per_step_runs"]))'
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi --no-roofline"
for prio in 0 1 0 1; do for wl in pds c4; do echo "$wl TLPK_SIDE_PRIO=$prio: $(TLPK_SIDE_PRIO=$prio timeout 300 python bench.py --workload $wl $B 2>/dev/null | python -c "$show")"; done; done | tee ${O}_side_prio.txt
for prio in 0 1; do echo "pds TLPK_GRAPH=0 TLPK_SIDE_PRIO=$prio: $(TLPK_GRAPH=0 TLPK_SIDE_PRIO=$prio timeout 300 python bench.py --workload pds $B 2>/dev/null | python -c "$show")"; done | tee -a ${O}_side_prio.txt
