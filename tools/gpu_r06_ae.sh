#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 1 0 1 0; do
echo "== TLPK_ANALYSE_POOL=$v"
TLPK_ANALYSE_POOL=$v timeout 600 python tools/analyse_phases.py 2>&1 | grep "setup (analyse" | tr '\n' ' '; echo
HEADLINE=1 TLPK_ANALYSE_POOL=$v timeout 600 python tools/analyse_phases.py 2>&1 | grep "setup (analyse" | tr '\n' ' '; echo
done
