#!/bin/bash
# Round 3, GPU session P: kernel trace of the default (concurrent) schedule of C4 for a timeline analysis (tools/timeline_overlap.py).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="--steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-headline --no-host-abi --no-small-lp --unpaired"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03_trace_c4 -- python bench.py $B > gpurun_out/r03_trace_c4.log 2>&1
f=$(ls gpurun_out/r03_trace_c4/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Stream_Id", "Queue_Id", "Grid_Size", "Workgroup_Size"]
keep = [k for k in keep if k in rows[0]]
with open("gpurun_out/r03_trace_c4_kernels.csv", "w", newline="") as g:
    w = csv.writer(g); w.writerow(keep)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void tlpk::", "").replace("tlpk::", "")
        w.writerow([name] + [r[k] for k in keep[1:]])
print(len(rows), "kernels", keep)
PY
rm -rf gpurun_out/r03_trace_c4
ls -la gpurun_out/r03_trace_c4_kernels.csv
