#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in good nofetch; do
cp tools/ab/libtlpk_$v.so tulip.jl_amd/libtlpk.so
echo "== $v"
timeout 300 python tools/chain_trace.py 2>&1 | grep -E "trsm  |k0= 6[0-9][0-9][0-9]|k0= 7" | cut -c1-230
done
cp tools/ab/libtlpk_good.so tulip.jl_amd/libtlpk.so
