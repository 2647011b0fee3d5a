#!/bin/bash
# Round 4, GPU session G: multi-device handles -- shard threads, reduce-scatter reduction, resident refinement; block detection on device.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_k2.py tests/test_hsd_device.py tests/test_blocks.py tests/test_abi.py -m gpu -q -x -s -k "multi or shard or refinement or blocks or abi or plain_c or device_loops" 2>&1 | grep -v "^$" | tail -14 | cut -c1-300
