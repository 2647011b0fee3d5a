#!/bin/bash
# Round 5, GPU session A: the whole GPU suite at the new defaults (threaded staging of the host-pointer calls, guarded refinement,
# one solve schedule for all stream groups, permuted CSR for k_rhs, automatic look-ahead), the default bench line, and A/B runs of
# the three knobs on the workloads they target.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05a
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -m1 "model name" /proc/cpuinfo ) > ${O}_host.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > ${O}_pytest.txt
tail -3 ${O}_pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05a_bench.json').read().strip().splitlines()[-1])
h=d.get('headline',{})
print('c4 %.2f ms runs %s host_abi %.2f unpaired %.2f frac %.3f solve %.3f ms frac %.3f pair/single %.3f' % (d['ms_per_step'], d['ms_per_step_runs'], d['host_abi']['ms_per_step'], d['unpaired_ms_per_step'], d['roofline']['frac'], d['solve_roofline']['ms_per_solve'], d['solve_roofline']['frac'], d['solve_roofline']['pair']['ms_over_single']))
print('headline %.2f ms runs %s host_abi %.2f unpaired %s frac %.3f solve frac %.3f' % (h['ms_per_step'], h.get('ms_per_step_runs'), h['host_abi']['ms_per_step'], h.get('unpaired_ms_per_step'), h['roofline']['frac'], h['solve_roofline']['frac']))
print('small', {k:(v.get('ms_per_step'), v.get('cpu_ms_per_step')) for k,v in d['small_lp'].items()}, 'c3', d['c3'].get('ms_per_step'))
print('kernel_ms', d['kernel_ms'])
PY
B="--steps 10 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get("kernel_ms",{}); s=d.get("solve_roofline",{})
print("%.2f ms  runs %s  host_abi %s  unpaired %s  solve %.3f ms pair/single %s  potrf %s update %s" % (d["ms_per_step"], d["ms_per_step_runs"], d.get("host_abi",{}).get("ms_per_step"), d.get("unpaired_ms_per_step"), s.get("ms_per_solve",0), s.get("pair",{}).get("ms_over_single"), k.get("potrf"), k.get("update")))'
for g in 1 0; do echo "c4 TLPK_SOLVE_ONE_GROUP=$g: $(TLPK_SOLVE_ONE_GROUP=$g timeout 300 python bench.py $B 2>/dev/null | python -c "$show")"; done
for g in 1 0; do echo "headline TLPK_SOLVE_ONE_GROUP=$g: $(TLPK_SOLVE_ONE_GROUP=$g timeout 400 python bench.py --workload headline --steps 5 --warmup 2 --no-cpu-baseline --no-small-lp --no-headline --no-c3 2>/dev/null | python -c "$show")"; done
for t in 4 0 8; do echo "c4 TLPK_COPY_THREADS=$t: $(TLPK_COPY_THREADS=$t timeout 300 python bench.py $B --no-roofline 2>/dev/null | python -c "$show")"; done
echo "c4 TLPK_COPY_NT=0: $(TLPK_COPY_NT=0 timeout 300 python bench.py $B --no-roofline 2>/dev/null | python -c "$show")"
for la in auto 0; do
  if [ $la = auto ]; then unset TLPK_LOOKAHEAD; else export TLPK_LOOKAHEAD=$la; fi
  echo "pds lookahead=$la: $(timeout 300 python bench.py --workload pds --steps 20 --warmup 3 --no-cpu-baseline --no-small-lp --no-headline --no-c3 --no-host-abi 2>/dev/null | python -c "$show")"
  echo "rank-local lookahead=$la: $(NLIST=8 timeout 200 python tools/rank_local_timing.py 2>&1 | grep nranks)"
done
unset TLPK_LOOKAHEAD
ONLY=refine NB=100 timeout 400 python tools/mpc_levers.py 2>&1 | tail -3
