cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hsd_device.py -m gpu -q -k "multi_device_handle" -s 2>&1 | grep -v "^$" | grep "one device\|assert\|Error\|passed\|failed" | cut -c1-700 > gpurun_out/r05h2_mode3.txt
TLPK_POTRF_MODE=2 timeout 600 python -m pytest tests/test_hsd_device.py -m gpu -q -k "multi_device_handle" -s 2>&1 | grep "one device\|assert\|Error\|passed\|failed" | cut -c1-700 > gpurun_out/r05h2_mode2.txt
cat gpurun_out/r05h2_mode3.txt; echo ----; cat gpurun_out/r05h2_mode2.txt
