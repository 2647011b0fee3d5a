#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2; do timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -6; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
