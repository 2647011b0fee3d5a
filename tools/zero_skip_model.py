"""Analyse-time model (no GPU): how many of k_update's executed flops multiply structural zeros of amalgamated supernodes,
and how many a per-(tile, K slab) skip list would remove.  Works on ONE diagonal block + the linking rows of a block-angular
LP of the benchmark shapes (the other blocks are statistically identical).

    python tools/zero_skip_model.py c4|headline [slab]
"""
import sys
import numpy as np
sys.path.insert(0, ".")
import tulip_jl_amd as tk
from workloads import block_angular_lp

TILE = 128
import os
REORDER = int(os.environ.get('REORDER', '1'))
LEX = int(os.environ.get('LEX', '0'))
CONTIG = int(os.environ.get('CONTIG', '1'))
NB_OUT = 256


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    slab = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    if which == "c4":
        A, rb = block_angular_lp(nblocks=2, mk=5000, nk=10000, m0=1000)
    else:
        A, rb = block_angular_lp(nblocks=2, mk=20000, nk=10000, m0=1000, ineq=True)
    kkt = tk.setup(A, tk.K1(), tk.Backend(device=-1, row_block=rb))
    m = A.shape[0]
    sym = kkt.symbolic
    parent = sym("etree"); Sp = sym("s_colptr"); Si = sym("s_rowidx"); cc = sym("colcount")
    f_f = sym("front_f"); f_ns = sym("front_ns"); f_col0 = sym("front_col0"); f_rowoff = sym("front_rowoff")
    f_block = sym("front_block"); rowidx = sym("rowidx")
    # true structure of L, columns of block 0 and the linking rows only (dense boolean)
    fb0 = [s for s in range(len(f_f)) if f_block[s] == 0]
    cols0 = np.concatenate([np.arange(f_col0[s], f_col0[s] + f_ns[s]) for s in fb0])
    cols0 = np.sort(cols0)
    n0 = cols0.size
    nlink = int((rb < 0).sum())
    first_link = m - nlink
    nloc = n0 + nlink
    lmap = np.full(m, -1, dtype=np.int64)
    lmap[cols0] = np.arange(n0)
    lmap[first_link:] = n0 + np.arange(nlink)

    def loc(r):      # permuted row -> local index (block-0 rows then linking rows)
        o = lmap[np.asarray(r)]
        assert (o >= 0).all()
        return o
    Lb = np.zeros((n0, nloc), dtype=bool)       # Lb[local col, local row]
    kids = [[] for _ in range(n0)]
    for j in cols0:
        p = parent[j]
        if p != -1 and p < first_link:
            kids[lmap[p]].append(lmap[j])
    for j in cols0:
        lj = lmap[j]
        row = Lb[lj]
        row[loc(Si[Sp[j]:Sp[j + 1]])] = True
        for c in kids[lj]:
            row |= Lb[c]
        row[:lj] = False
        assert row.sum() == cc[j], (j, row.sum(), cc[j])
    tot_alg = tot_exec = tot_skip = tot_skip_wave = 0.0
    big = []
    for s in fb0:
        f, ns, col0 = int(f_f[s]), int(f_ns[s]), int(f_col0[s])
        if ns < 32 or f < 256:
            # small fronts: count them as executed = padded (no skipping), they hardly matter
            pass
        rows = loc(rowidx[f_rowoff[s]:f_rowoff[s] + f])
        M = Lb[lmap[col0]:lmap[col0] + ns][:, rows].T
        assert lmap[col0 + ns - 1] == lmap[col0] + ns - 1          # M[row, col] true structure, f x ns
        # algorithmic flops of k_update (same convention as symbolic.cpp: flops_update_alg)
        alg = 0.0
        for c in range(ns):
            r = min((c // NB_OUT + 1) * NB_OUT, ns) - c
            l = cc[col0 + c] - r
            if l > 0:
                alg += float(l) * l
        if REORDER and ns >= 256:
            # dense tail [t, ns): columns whose padding is small; its pivots are re-ordered by the first prefix column in which
            # the row is a true nonzero (rows that no prefix column touches: t)
            pad = (f - np.arange(ns)) - cc[col0:col0 + ns]
            bad = np.nonzero(pad > 0.0625 * (f - np.arange(ns)))[0]
            t = int(bad.max()) + 1 if bad.size else 0
            t = (t + 15) // 16 * 16
            if 0 < t < ns:
                first = np.where(M[:, :t].any(axis=1), M[:, :t].argmax(axis=1), t)      # e(r), capped at t
                order = np.arange(f)
                tail = np.arange(t, ns)
                if LEX:
                    nps = t // slab
                    bits = np.stack([M[t:ns, k * slab:(k + 1) * slab].any(axis=1) for k in range(nps)], axis=0)   # [slab, row]
                    # np.lexsort: last key is the primary one; primary = last prefix slab, rows WITH a nonzero first
                    order[t:ns] = tail[np.lexsort(tuple(~bits[k] for k in range(nps)))]
                else:
                    order[t:ns] = tail[np.argsort(first[t:ns], kind="stable")]
                M = M[order]            # rows permuted; the tail COLUMNS are permuted alike but they are treated as dense below
                M[:, t:] = np.tril(np.ones((f, ns), dtype=bool))[:, t:]
                print(f"   front {s}: dense tail starts at column {t} of {ns}")
        ntr = (f + TILE - 1) // TILE
        nsl = (ns + slab - 1) // slab
        # flag[t, k] = row tile t has a structural nonzero in slab k
        flag = np.zeros((ntr, nsl), dtype=bool)
        for t in range(ntr):
            blk = M[t * TILE:(t + 1) * TILE]
            for k in range(nsl):
                flag[t, k] = blk[:, k * slab:(k + 1) * slab].any()
        if ns >= 256 and os.environ.get('DUMP'):
            for k in range(0, min(nsl, 48), 2):
                print("   slab", k, "".join("#" if flag[t, k] else "." for t in range(ntr)), "true nnz in slab cols:", int(M[:, k*slab:(k+1)*slab].sum()))
        ex = sk = 0.0
        # block columns [ko, ko + 256): targets columns J tiles in the block column, rows I >= J; K = [0, ko)
        targets = []
        for ko in range(NB_OUT, ns, NB_OUT):
            jl = min(ko + NB_OUT, ns)
            for j0 in range(ko, jl, TILE):
                for i0 in range(j0, f, TILE):
                    targets.append((i0 // TILE, j0 // TILE, ko))
        for j0 in range(ns, f, TILE):       # update matrix: K = [0, ns); tiles relative to ns in the real kernel, close enough
            for i0 in range(j0, f, TILE):
                targets.append((i0 // TILE, j0 // TILE, ns))
        for (ti, tj, kend) in targets:
            nk = (kend + slab - 1) // slab
            ex += 2.0 * TILE * TILE * kend
            both = flag[ti, :nk] & flag[tj, :nk]
            if CONTIG:
                nzs = np.nonzero(both)[0]
                sk += 2.0 * TILE * TILE * (kend - slab * nzs[0]) if nzs.size else 0.0
            else:
                sk += 2.0 * TILE * TILE * min(kend, slab * both.sum())
        tot_alg += alg; tot_exec += ex; tot_skip += sk
        if ns >= 256:
            big.append((s, f, ns, alg, ex, sk, flag.mean()))
    print(f"{which}: block-0 fronts {len(fb0)}, slab {slab}")
    for b in big:
        print("  front %d f=%d ns=%d alg=%.3e exec=%.3e (x%.3f) skip=%.3e (x%.3f) nonzero (tile,slab) fraction %.3f" %
              (b[0], b[1], b[2], b[3], b[4], b[4] / b[3], b[5], b[5] / b[3], b[6]))
    print(f"  total alg {tot_alg:.4e} exec(full tiles) {tot_exec:.4e} = x{tot_exec / tot_alg:.3f}; with skip lists {tot_skip:.4e} = x{tot_skip / tot_alg:.3f}")


main()
