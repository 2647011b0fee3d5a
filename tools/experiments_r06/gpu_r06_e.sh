#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bump_replay.py tests/test_hsd_device.py -m gpu -x -q -s -k "replays or resident_refinement" > gpurun_out/r06e_pytest.txt 2>&1
tail -30 gpurun_out/r06e_pytest.txt
