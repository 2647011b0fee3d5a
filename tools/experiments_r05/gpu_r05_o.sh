#!/bin/bash
# Round 5, GPU session O: the whole GPU suite + smoke at the round's last code state
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r05_final_pytest.txt 2>&1
tail -3 gpurun_out/r05_final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_final_smoke.txt 2>&1; tail -1 gpurun_out/r05_final_smoke.txt
