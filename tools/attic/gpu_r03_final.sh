#!/bin/bash
# Round 3, final GPU session: the whole -m gpu suite, smoke(), then the measurement artifacts of tools/collect_profiles_r03.sh.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_final_pytest.txt 2>&1
tail -5 gpurun_out/r03_final_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_final_smoke.txt 2>&1; tail -2 gpurun_out/r03_final_smoke.txt
bash tools/collect_profiles_r03.sh
