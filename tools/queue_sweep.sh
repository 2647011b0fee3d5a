run() { echo "== $*"; env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],1))
"; }
run TLPK_STREAMS=2 GPU_MAX_HW_QUEUES=8
run TLPK_STREAMS=3 GPU_MAX_HW_QUEUES=8
run TLPK_STREAMS=4 GPU_MAX_HW_QUEUES=8
run TLPK_STREAMS=4 GPU_MAX_HW_QUEUES=12
run TLPK_STREAMS=8 GPU_MAX_HW_QUEUES=16
