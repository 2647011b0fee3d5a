#!/bin/bash
# Round 3, GPU session N: the pair of solves in split-phase form (sharded handles, multi-device HSD loop, bench.py's collective branch).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --maxfail=6 -k "two_right_hand or sharded_split or multi_device or force or collectives or bench" > gpurun_out/r03_n_pytest.txt 2>&1
tail -12 gpurun_out/r03_n_pytest.txt
NSHARDS=2 timeout 300 python tools/solve_c4_lp.py 2>/dev/null | grep -A1 "HSD"
python bench.py --steps 10 --warmup 3 --force-collectives --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('forced collectives at N=1, paired: ms/step %.3f' % d['ms_per_step'], d['config'].get('solve_schedule'), d.get('residuals'))"
python bench.py --steps 10 --warmup 3 --force-collectives --no-cpu-baseline --unpaired 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('forced collectives at N=1, unpaired: ms/step %.3f' % d['ms_per_step'])"
