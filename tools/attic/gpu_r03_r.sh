#!/bin/bash
# Round 3, GPU session R: measurement artifacts with packed panels (tools/collect_profiles_r03.sh), then the C3 shape at 1e5 and
# 2e5 rows (the second one only fits with packed panels: 151 GB against 301 GB).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_final_pytest.txt 2>&1; tail -3 gpurun_out/r03_final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_final_smoke.txt 2>&1; tail -1 gpurun_out/r03_final_smoke.txt
bash tools/collect_profiles_r03.sh
timeout 600 python bench.py --workload c3 --c3-rows 100000 --steps 2 --warmup 1 --no-host-abi --no-cpu-baseline > gpurun_out/r03_c3_100k_rows_bench.json 2> gpurun_out/r03_c3_100k.err
timeout 900 python bench.py --workload c3 --c3-rows 200000 --steps 1 --warmup 0 --no-host-abi --no-cpu-baseline --no-roofline > gpurun_out/r03_c3_200k_rows_bench.json 2> gpurun_out/r03_c3_200k.err
tail -3 gpurun_out/r03_c3_200k.err
python - <<'PY'
import json
for f in ("r03_final_bench.json", "r03_c3_50k_rows_bench.json", "r03_c3_100k_rows_bench.json", "r03_c3_200k_rows_bench.json"):
    try:
        d = json.load(open("gpurun_out/" + f)); c = d["config"]
        print(f, "ms/step %.1f" % d["ms_per_step"], "stored/nnzL %.3f" % (c["nnzL_stored"] / max(c["nnzL"], 1)), "device GB %.1f" % (c.get("device_bytes", 0) / 1e9), d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
