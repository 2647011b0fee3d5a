"""Round 6 diagnostics: time line of the dependency-driven factorisation (k_chain) of the pds-class LP (or C4 with WL=c4) from the per-item time stamps the
kernel writes with TLPK_CHAIN_TRACE=1: for the largest chain launch, per role: items, time waiting for counters, time working, time publishing; and the
critical path of the top front block column by block column (diagonal tiles -> diagonal block -> first strip)."""
import os, sys
import numpy as np
os.environ["TLPK_CHAIN_TRACE"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tulip_jl_amd as tk
from tulip_jl_amd.problem import standard_form
from lp_generators import multicommodity_lp
from helpers import ipm_like_data

wl = os.environ.get("WL", "pds")
if wl == "pds":
    A = standard_form(multicommodity_lp()).A; rb = None
else:
    from workloads import block_angular_lp
    A, rb = block_angular_lp(nblocks=int(os.environ.get("BLOCKS", "8")))[:2]
m, n = A.shape
kkt = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb))
th, rp, rd, xp, xd = ipm_like_data(m, n, 1)
for _ in range(3):
    try:
        tk.update(kkt, th, rp, rd)
    except Exception as e:      # (experiments that break the numbers still have a time line)
        print("update failed:", str(e)[:120])
items = kkt.symbolic("chain_items").reshape(-1, 12)
L = kkt.symbolic("factor_launches").reshape(-1, 3)
tr = kkt.symbolic("chain_trace").reshape(-1, 4).astype(np.float64) / 100.0      # microseconds (100 MHz)
ut = kkt.symbolic("update_tasks").reshape(-1, 10); pt = kkt.symbolic("potrf_tasks").reshape(-1, 4); tt = kkt.symbolic("trsm_tasks").reshape(-1, 6)
names = ["update", "potrf", "trsm", "reduce"]
for kind, first, count in L:
    if kind != 22:
        continue
    it = items[first:first + count]; t = tr[first:first + count]
    t0 = t[:, 0].min()
    print(f"chain launch: {count} items, span {t[:, 3].max() - t0:.1f} us")
    for r in range(4):
        sel = it[:, 0] == r
        if sel.any():
            w = t[sel, 1] - t[sel, 0]; x = t[sel, 2] - t[sel, 1]; p = t[sel, 3] - t[sel, 2]
            print(f"  {names[r]:7s} {sel.sum():6d} items: wait mean {w.mean():8.1f} max {w.max():8.1f} | work mean {x.mean():7.1f} max {x.max():7.1f} sum {x.sum()/1e3:8.2f} ms | publish mean {p.mean():5.2f} max {p.max():5.2f}")
    if count < 500:
        continue
    # critical path per block column of the biggest front
    pf = np.nonzero(it[:, 0] == 1)[0]
    fronts = pt[it[pf, 1], 0]; big = np.bincount(fronts).argmax()
    prev_done = None
    print("  block column: diag tiles (first ready -> last done) | potrf wait-after-tiles, work | first strip: released (relative to the potrf's publish: negative = early entry, "
          "the strip waits inside its role, TLPK_CHAIN_EARLY), time in its role | last strip published, and how long after the potrf | period")
    for q in pf:
        if pt[it[q, 1], 0] != big:
            continue
        k0 = pt[it[q, 1], 1]
        # diagonal-tile adders of this block column: update / reduce items that signal one of the potrf's wait counters
        ws = {it[q, 3] if it[q, 4] else -1, it[q, 6] if it[q, 7] else -1, it[q, 9]} - {-1}
        dsel = np.isin(it[:, 11], list(ws)) & (it[:, 0] != 1)
        strips = np.nonzero((it[:, 0] == 2) & (tt[it[:, 1] % len(tt), 0] == big) & (tt[it[:, 1] % len(tt), 1] == k0))[0]
        s0 = strips[0] if len(strips) else None
        d_first = t[dsel, 1].min() - t0 if dsel.any() else float("nan"); d_last = t[dsel, 3].max() - t0 if dsel.any() else float("nan")
        line = f"  k0={k0:5d}: tiles {d_first:8.1f} -> {d_last:8.1f} | potrf ready {t[q,1]-t0:8.1f} (+{t[q,1]-t0-d_last:5.1f}) work {t[q,2]-t[q,1]:6.1f} pub {t[q,3]-t[q,2]:4.1f}"
        if s0 is not None:
            line += f" | strip ready {t[s0,1]-t0:8.1f} ({t[s0,1]-t[q,3]:+6.1f}) work {t[s0,2]-t[s0,1]:5.1f}; last strip done {t[strips,3].max()-t0:8.1f} ({t[strips,3].max()-t[q,3]:+5.1f})"
        if prev_done is not None:
            line += f" | period {t[q,3]-t0-prev_done:6.1f}"
        prev_done = t[q, 3] - t0
        print(line)
