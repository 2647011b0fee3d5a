#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r06f
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_hsd_device.py -m gpu -x -q -k "two_right_hand or pair or host_pointer or paired" > ${O}_pytest.txt 2>&1
tail -4 ${O}_pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-host-abi --no-small-lp --no-c3 > ${O}_bench_c4.json 2> ${O}_bench_c4.err
python - <<P
import json
d=json.load(open("${O}_bench_c4.json")); print("c4", d["ms_per_step"], d["solve_roofline"]["ms_per_solve"], d["solve_roofline"]["frac"], d["solve_roofline"]["pair"], d.get("unpaired_ms_per_step"))
print(d["kernel_ms"]); print(d["roofline"]["frac"], d["roofline"].get("chain_launches"))
P
timeout 600 python bench.py --workload headline --steps 4 --warmup 1 --no-cpu-baseline --no-host-abi > ${O}_bench_headline.json 2> ${O}_bench_headline.err
python - <<P
import json
d=json.load(open("${O}_bench_headline.json")); print("headline", d["ms_per_step"], d["solve_roofline"]["ms_per_solve"], d["solve_roofline"]["frac"], d["solve_roofline"]["pair"], d.get("unpaired_ms_per_step"))
P
