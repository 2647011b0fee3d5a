"""Slot simulation of the k_update launches (no GPU): 512 resident workgroups (2 per CU) draw the tiles of a launch in list order;
duration of a tile ~ its executed K slabs + a fixed prologue / epilogue cost.  Compares list order with longest-first order.
    python tools/update_makespan.py c4|headline [nblocks]"""
import sys, heapq
import numpy as np
sys.path.insert(0, ".")
import tulip_jl_amd as tk
from workloads import block_angular_lp

which = sys.argv[1]; nb = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if which == "c4" else 100)
A, rb = (block_angular_lp(nblocks=nb) if which == "c4" else block_angular_lp(nblocks=nb, mk=20000, nk=10000, m0=1000, ineq=True))
kkt = tk.setup(A, tk.K1(), tk.Backend(device=-1, row_block=rb, streams=1))
ut = kkt.symbolic("update_tasks").reshape(-1, 10)
fl = kkt.symbolic("factor_launches").reshape(-1, 3)
OVERHEAD = 3.0       # slabs-equivalent of prologue + epilogue (~14 us vs ~4.3 us per slab round)
SLOTS = 512


def makespan(d):
    h = [0.0] * min(SLOTS, len(d))
    heapq.heapify(h)
    for x in d:
        heapq.heappush(h, heapq.heappop(h) + x)
    return max(h)


tot = {"list": 0.0, "lpt": 0.0, "ideal": 0.0}
for kind, first, count in fl:
    if kind != 3:
        continue
    t = ut[first:first + count]
    slabs = np.where(t[:, 8] > 0, t[:, 9] + (t[:, 2] % 16 > 0), (t[:, 2] + 15) // 16).astype(float) + OVERHEAD
    tot["list"] += makespan(slabs); tot["lpt"] += makespan(np.sort(slabs)[::-1]); tot["ideal"] += slabs.sum() / SLOTS
print(which, nb, "blocks: sum over launches of the simulated makespan (slab rounds): list order %.0f, longest first %.0f, perfect packing %.0f" %
      (tot["list"], tot["lpt"], tot["ideal"]))
