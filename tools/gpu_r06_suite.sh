#!/bin/bash
# the whole GPU suite + smoke at the current code state -> gpurun_out/r06_final_pytest.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r06_final_pytest.txt 2>&1
tail -12 gpurun_out/r06_final_pytest.txt
