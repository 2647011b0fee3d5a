#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
timeout 600 python tools/analyse_phases.py 2>&1 | grep "setup (analyse" | awk "{printf \"%s \", \$6}"
HEADLINE=1 timeout 600 python tools/analyse_phases.py 2>&1 | grep "setup (analyse" | awk "{printf \"%s \", \$6}"; echo
done
HEADLINE=1 TLPK_TIMING=1 timeout 600 python tools/analyse_phases.py 2>&1 | tail -33 | head -31
