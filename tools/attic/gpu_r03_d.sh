#!/bin/bash
# Round 3, GPU session D: new paths (paired solves, graph replay, K2 multi-device) + graph on/off timings on the small and the bench LPs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -k "two_right_hand or graph_replay or paired_h_system or k2_single_process or multi_device or detected_blocks or auto_blocks or bench_multi or bench_small" > gpurun_out/r03_d_pytest.txt 2>&1
tail -25 gpurun_out/r03_d_pytest.txt
B="--no-headline --no-cpu-baseline --no-small-lp --no-roofline"
{
for wl in stair25 pds; do for g in 0 1; do
  echo "== workload $wl TLPK_GRAPH=$g"
  TLPK_GRAPH=$g timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 $B 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ms_per_step %.4f  host_abi %.4f  unpaired %.4f  launches %d + 4 x %d' % (d['ms_per_step'], d['host_abi']['ms_per_step'], d.get('unpaired_ms_per_step', float('nan')), d['config']['launches_update'], d['config']['launches_solve']))"
done; done
for g in 0 1; do
  echo "== workload c4 TLPK_GRAPH=$g"
  TLPK_GRAPH=$g timeout 300 python bench.py --steps 20 --warmup 5 $B 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ms_per_step %.4f  host_abi %.4f  unpaired %.4f' % (d['ms_per_step'], d['host_abi']['ms_per_step'], d.get('unpaired_ms_per_step', float('nan'))))"
done
} > gpurun_out/r03_small_lp_graph.txt 2>&1
cat gpurun_out/r03_small_lp_graph.txt
