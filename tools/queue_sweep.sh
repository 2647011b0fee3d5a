run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],1))
"; }
run TLPK_STREAMS=1
run TLPK_STREAMS=2
