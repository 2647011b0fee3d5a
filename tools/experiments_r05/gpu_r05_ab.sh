#!/bin/bash
# Round 5, session AB: KKT.setup after the host-side changes (no zero-fill of the assembly arrays, defaults / cursor / CSR transposition on the host threads, no compaction
# copies in upload_all when every entry is local): analyse phases, end-to-end LPs (setup line), then the whole GPU suite + smoke.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TLPK_TIMING=1 timeout 300 python tools/analyse_phases.py 2>&1 | grep "assembly\|copy A\|setup" | tail -4 | tee gpurun_out/r05ab_analyse.txt
HEADLINE=1 TLPK_TIMING=1 timeout 300 python tools/analyse_phases.py 2>&1 | grep "assembly\|copy A\|setup" | tail -4 | tee -a gpurun_out/r05ab_analyse.txt
ALGS=HSD timeout 300 python tools/solve_c4_lp.py 2>&1 | tail -2 | tee gpurun_out/r05ab_lp.txt
HEADLINE=1 ALGS=HSD timeout 300 python tools/solve_c4_lp.py 2>&1 | tail -2 | tee -a gpurun_out/r05ab_lp.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r05_final_pytest.txt 2>&1
tail -3 gpurun_out/r05_final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_final_smoke.txt 2>&1; tail -1 gpurun_out/r05_final_smoke.txt
