run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],2))
"; }
run TLPK_MACRO_TILES=0
run TLPK_MACRO_TILES=2048
run TLPK_MACRO_TILES=0
run TLPK_MACRO_TILES=2048
run TLPK_MACRO_TILES=3000
