"""Could the solve sweeps skip the amalgamation padding of the panels (stored / nnz(L) = 1.14 on C4, 1.24 on the north-star LP)?  Share of the lower trapezoids of the fronts with
structure bits (analyse step 13c: one bit per 16-column slab x 16-row group) that is structurally zero at the granularities a sweep could skip: 16 x 16 cells, 16-row x 64-column
strips, 64 x 64 blocks (a wave of a sweep item covers 64 rows of a 64-column block).  Host analyse only (2 blocks of the shape; the share per block is that of the LP)."""
import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from workloads import block_angular_lp
from test_symbolic import analyse_only
HEAD = os.environ.get("HEADLINE") == "1"
A, rb = (block_angular_lp(2, 20000, 10000, 1000, 4, 0.5, ineq=True) if HEAD else block_angular_lp(nblocks=2))
kkt = analyse_only(A, row_block=rb)
so = kkt.symbolic("skip_off"); sb = kkt.symbolic("skip_bits").view(np.uint64)
f_f, f_ns = kkt.symbolic("front_f"), kkt.symbolic("front_ns")
tot = z16 = z16x64 = z64 = 0; tot_all = 0
for sf in range(len(f_f)):
    f, ns = int(f_f[sf]), int(f_ns[sf])
    tot_all += f * ns - ns * (ns - 1) // 2
    if so[sf] < 0 or f < 256: continue
    nsl, ng = (ns + 15) // 16, (f + 15) // 16
    W = (ng + 63) // 64
    b = sb[so[sf]: so[sf] + nsl * W].reshape(nsl, W)
    words = np.repeat(b, 64, axis=1)[:, :ng]
    nz = ((words >> (np.arange(ng, dtype=np.uint64) & np.uint64(63))) & np.uint64(1)).astype(bool)      # [slab][row group] True = nonzero
    for sl in range(nsl):
        g0 = sl                       # rows from the slab's own diagonal group down
        cells = nz[sl, g0:]
        tot += cells.size; z16 += (~cells).sum()
    # 64-col blocks x 64-row groups
    for cb in range((nsl + 3) // 4):
        sls = range(4 * cb, min(4 * cb + 4, nsl))
        for rg in range(cb, (ng + 3) // 4):
            sub = nz[sls.start:sls.stop, 4 * rg: min(4 * rg + 4, ng)]
            if not sub.any(): z64 += sub.size
        for g in range(4 * cb, ng):
            sub = nz[sls.start:sls.stop, g]
            if not sub.any(): z16x64 += sub.size
print("fronts with bits: 16x16 cells in the lower trapezoid %d; structurally zero: %.1f %% at 16x16, %.1f %% as 16-row x 64-col strips, %.1f %% as 64x64 blocks" % (tot, 100*z16/tot, 100*z16x64/tot, 100*z64/tot))
st = kkt.stats(); print("stored/nnzL", st["nnzL_stored"]/st["nnzL"])
