"""Analyse-only statistics of the amalgamated supernodal tree (no GPU): padded flops, extend-add volume, fronts, levels, launches.
    python tools/amalg_stats.py c4|headline [nblocks]      (env TLPK_RELAX_* knobs apply)"""
import sys
import numpy as np
sys.path.insert(0, ".")
import tulip_jl_amd as tk
from workloads import block_angular_lp

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if which == "c4":
    A, rb = block_angular_lp(nblocks=nb, mk=5000, nk=10000, m0=1000)
else:
    A, rb = block_angular_lp(nblocks=nb, mk=20000, nk=10000, m0=1000, ineq=True)
kkt = tk.setup(A, tk.K1(), tk.Backend(device=-1, row_block=rb))
st = kkt.stats()
sym = kkt.symbolic
f = sym("front_f"); ns = sym("front_ns"); par = sym("front_parent"); single = sym("front_single")
rs = (f - ns).astype(float)
ea = float((rs[par >= 0] ** 2).sum()) / 2
big = f >= 2048
fl = sym("factor_launches").reshape(-1, 3)
print(f"{which} x{nb}: fronts {st['n_supernodes']} (single {int(single.sum())}, big {int(big.sum())}) levels {st['n_levels']} "
      f"exec/alg {st['flops_update'] / st['flops_update_alg']:.3f} flops_update {st['flops_update']:.3e} alg {st['flops_update_alg']:.3e} "
      f"stored/nnzL {st['nnzL_stored'] / st['nnzL']:.3f} extend-add entries {ea:.3e} ({24 * ea / 1e9:.2f} GB) "
      f"children of big fronts {int(np.isin(par, np.nonzero(big)[0]).sum())} launches {len(fl)} update tasks {len(sym('update_tasks')) // 10}")
