"""Seeded LP generators standing in for the BASELINE.json configs whose .mps files are not in the
image (no network): configs[1] "Netlib 25fv47" and configs[4] "pds-20 or equivalent".

Test infrastructure.  Pure numpy/scipy (PCG64 streams are stable across numpy versions); nothing
from the reference.  Both generators build LPs that are primal AND dual feasible by construction
(a primal point inside all bounds; c = A'y* + z* with multiplier signs that match the row / bound
types), so an optimum exists and HiGHS (scipy.optimize.linprog) gives an independent optimal value.

  staircase_lp      "25fv47-class": ~820 rows x ~1570 columns, ~10 000 nonzeros, staircase
                    (multi-period) structure, mixed E / L / G / range rows, lower-bounded, boxed and
                    free columns, coefficient magnitudes over several decades.  (25fv47 itself:
                    821 rows, 1571 columns, 10 400 nonzeros.)
  multicommodity_lp "PDS-class" (patient distribution system = multicommodity flow with joint arc
                    capacities): K commodities on a random layered digraph; per commodity one flow
                    conservation row per node (equalities), joint capacity rows on a subset of the
                    arcs (inequalities), per-commodity arc bounds on some arcs.  At the default size
                    m ~ 3.4e4 rows like pds-20 (33 874 rows, 105 728 columns, 230 200 nonzeros).
  write_free_mps    free-format MPS writer (what the reference reads through QPSReader,
                    /root/reference/src/Interfaces/tulip_julia_api.jl:18-39), so that both backends
                    are fed "the same .mps input".
"""
import numpy as np
import scipy.sparse as sp

from ipm_harness import LP

INF = float("inf")


def _dual_feasible_cost(A, lcon, ucon, lvar, uvar, rng):
    """c = A'y + z with y_i free on E rows, <= 0 on L rows, >= 0 on G rows, 0 on range rows;
    z_j >= 0 for lower-bounded, <= 0 for upper-bounded only, any sign for boxed, 0 for free columns."""
    m, n = A.shape
    y = rng.standard_normal(m)
    only_u = np.isinf(lcon) & np.isfinite(ucon)
    only_l = np.isfinite(lcon) & np.isinf(ucon)
    rng_row = np.isfinite(lcon) & np.isfinite(ucon) & (lcon < ucon)
    y[only_u] = -np.abs(y[only_u]); y[only_l] = np.abs(y[only_l]); y[rng_row] = 0.0
    z = rng.uniform(0.1, 1.0, n)
    fin_l, fin_u = np.isfinite(lvar), np.isfinite(uvar)
    z[~fin_l & fin_u] *= -1.0
    boxed = fin_l & fin_u
    z[boxed] *= rng.choice([-1.0, 1.0], size=int(boxed.sum()))
    z[~fin_l & ~fin_u] = 0.0
    return A.T @ y + z


def staircase_lp(seed=25047, periods=12, rows_per=68, cols_per=131, name="STAIR25"):
    """Netlib-25fv47-class staircase LP (see the module docstring)."""
    rng = np.random.default_rng(seed)
    T = periods
    m, n = T * rows_per + 5, T * cols_per
    ri, ci, vv = [], [], []
    for t in range(T):
        r0, c0 = t * rows_per, t * cols_per
        for j in range(cols_per):
            k = int(rng.integers(3, 8))                            # within-period entries
            rr = rng.choice(rows_per, size=k, replace=False)
            ri += list(r0 + rr); ci += [c0 + j] * k
            vv += list(rng.standard_normal(k) * 10.0 ** rng.integers(-2, 3, k))
            if t + 1 < T and rng.random() < 0.45:                  # carry-over into the next period
                k2 = int(rng.integers(1, 3))
                rr = rng.choice(rows_per, size=k2, replace=False)
                ri += list(r0 + rows_per + rr); ci += [c0 + j] * k2
                vv += list(rng.standard_normal(k2))
        # a few rows of the period are dense-ish budget rows
    for q in range(5):                                             # global rows over a column sample
        cols = rng.choice(n, size=60, replace=False)
        ri += [T * rows_per + q] * 60; ci += list(cols); vv += list(rng.uniform(0.5, 2.0, 60))
    A = sp.csc_matrix((vv, (ri, ci)), shape=(m, n)); A.sum_duplicates(); A.sort_indices()
    # every row needs an entry: add a unit entry on a random column to empty rows
    empty = np.nonzero(np.diff(A.tocsr().indptr) == 0)[0]
    if empty.size:
        A = (A + sp.csc_matrix((np.ones(empty.size), (empty, rng.integers(0, n, empty.size))), shape=(m, n))).tocsc()
        A.sort_indices()
    # column bounds: 80 % [0, inf), 14 % boxed [0, u], 4 % free, 2 % (-inf, u]
    kind = rng.choice(4, size=n, p=[0.80, 0.14, 0.04, 0.02])
    lvar = np.zeros(n); uvar = np.full(n, INF)
    uvar[kind == 1] = rng.uniform(1.0, 10.0, int((kind == 1).sum()))
    lvar[kind == 2] = -INF
    lvar[kind == 3] = -INF; uvar[kind == 3] = rng.uniform(0.0, 5.0, int((kind == 3).sum()))
    x0 = rng.uniform(0.2, 0.9, n)
    x0[kind == 1] *= uvar[kind == 1]
    x0[kind == 2] = rng.standard_normal(int((kind == 2).sum()))
    x0[kind == 3] = uvar[kind == 3] - rng.uniform(0.1, 1.0, int((kind == 3).sum()))
    act = A @ x0
    # row types: 55 % E, 25 % L, 15 % G, 5 % range
    rt = rng.choice(4, size=m, p=[0.55, 0.25, 0.15, 0.05])
    slack = rng.uniform(0.05, 1.0, m) * (1 + np.abs(act))
    lcon = act.copy(); ucon = act.copy()
    lcon[rt == 1] = -INF; ucon[rt == 1] = (act + slack)[rt == 1]
    ucon[rt == 2] = INF; lcon[rt == 2] = (act - slack)[rt == 2]
    lcon[rt == 3] = (act - slack)[rt == 3]; ucon[rt == 3] = (act + slack)[rt == 3]
    c = _dual_feasible_cost(A, lcon, ucon, lvar, uvar, rng)
    return LP(A, c, 0.0, lcon, ucon, lvar, uvar, True, name)


def multicommodity_lp(seed=2020, nodes=3080, arcs_per_node=3, K=10, cap_frac=0.33, bound_frac=0.2, name="PDSEQ20"):
    """PDS-class multicommodity flow LP (see the module docstring).  Default size: K*nodes = 30 800
    conservation rows + ~3 050 joint capacity rows = ~33 850 rows, K*arcs = 92 400 columns."""
    rng = np.random.default_rng(seed)
    E = nodes * arcs_per_node
    tail = np.repeat(np.arange(nodes), arcs_per_node)
    # layered digraph with mostly local arcs (like a transport network): head = tail + small offset
    off = rng.integers(1, 40, E)
    far = rng.random(E) < 0.05
    off[far] = rng.integers(1, nodes, int(far.sum()))
    head = (tail + off) % nodes
    n = K * E
    # node-arc incidence per commodity (flow out of tail: +1, into head: -1)
    Ninc = sp.csc_matrix((np.concatenate([np.ones(E), -np.ones(E)]),
                          (np.concatenate([tail, head]), np.concatenate([np.arange(E), np.arange(E)]))), shape=(nodes, E))
    cons = sp.block_diag([Ninc] * K, format="csc")
    capped = np.nonzero(rng.random(E) < cap_frac)[0]
    ncap = capped.size
    J = sp.csc_matrix((np.ones(ncap), (np.arange(ncap), capped)), shape=(ncap, E))
    A = sp.vstack([cons, sp.hstack([J] * K, format="csc")], format="csc")
    A.sort_indices()
    m = K * nodes + ncap
    # a feasible multicommodity flow: random nonnegative arc flows; supplies = their divergence
    x0 = rng.uniform(0.0, 2.0, n) * (rng.random(n) < 0.6)
    b_cons = cons @ x0
    joint = (sp.hstack([J] * K, format="csc") @ x0)
    lcon = np.concatenate([b_cons, np.full(ncap, -INF)])
    ucon = np.concatenate([b_cons, joint + rng.uniform(0.5, 3.0, ncap)])
    lvar = np.zeros(n); uvar = np.full(n, INF)
    bd = rng.random(n) < bound_frac
    uvar[bd] = x0[bd] + rng.uniform(0.5, 2.0, int(bd.sum()))
    cost = np.tile(rng.uniform(1.0, 10.0, E), K) * rng.uniform(0.8, 1.2, n)      # positive arc costs: bounded below
    return LP(A, cost, 0.0, lcon, ucon, lvar, uvar, True, name)


def bump_lp(seed=99, m=60, n=160, pairs=20, big=3.0e5):
    """An LP engineered so that the normal-equations Cholesky FAILS numerically late in the IPM run
    and the reference's retry loop (/root/reference/src/IPM/HSD/step.jl:35-51: regularisations x100,
    up to 3 times) has to fire: `pairs` free columns with entries `big` and `2*big` in two rows each.
    For a free column theta_inv = 0, so D_j = 1/regP_j grows to 1/sqrt(eps) = 6.7e7 and the two rows
    become numerically parallel in S = A D A' + Rd (rank-one term ~ 6e18 against O(1) remainders):
    the Schur complement of the second row is positive in exact arithmetic and rounding noise of
    size ~1e3 in fp64."""
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=4.0 / m, random_state=seed, format="lil", data_rvs=rng.standard_normal)
    for i in range(m):
        A[i, rng.integers(0, n)] = 1.0
    free_cols = []
    for q in range(pairs):
        j = n + q
        free_cols.append(j)
    B = sp.lil_matrix((m, pairs))
    rows = rng.permutation(m)[: 2 * pairs].reshape(pairs, 2)
    for q in range(pairs):
        B[rows[q, 0], q] = big; B[rows[q, 1], q] = 2.0 * big
    A = sp.hstack([A.tocsc(), B.tocsc()], format="csc"); A.sort_indices()
    ntot = n + pairs
    lvar = np.zeros(ntot); uvar = np.full(ntot, INF)
    lvar[n:] = -INF
    x0 = np.concatenate([rng.uniform(0.5, 1.5, n), rng.standard_normal(pairs) * 1e-3])
    b = A @ x0
    c = _dual_feasible_cost(A, b, b, lvar, uvar, rng)
    return LP(A, c, 0.0, b, b, lvar, uvar, True, "BUMP")


def write_free_mps(lp, path):
    """Free-format MPS (row names R<i>, column names C<j>, objective row COST).  Values are written
    with repr() so that reading the file back reproduces the LP bit for bit."""
    A = lp.A.tocsc(); m, n = A.shape
    with open(path, "w") as fh:
        fh.write(f"NAME {lp.name or 'LP'}\n")
        if not lp.objsense_min:
            fh.write("OBJSENSE\n    MAX\n")
        fh.write("ROWS\n N COST\n")
        rtype = []
        for i in range(m):
            lb, ub = lp.lcon[i], lp.ucon[i]
            t = "E" if lb == ub else ("L" if np.isinf(lb) and np.isfinite(ub) else ("G" if np.isfinite(lb) and np.isinf(ub) else "R"))
            if t == "R" and np.isinf(lb) and np.isinf(ub):
                raise ValueError("free rows are not written")
            rtype.append(t)
            fh.write(f" {'L' if t == 'R' else t} R{i}\n")
        fh.write("COLUMNS\n")
        for j in range(n):
            if lp.obj[j] != 0.0:
                fh.write(f" C{j} COST {float(lp.obj[j])!r}\n")
            for p in range(A.indptr[j], A.indptr[j + 1]):
                fh.write(f" C{j} R{A.indices[p]} {float(A.data[p])!r}\n")
            if lp.obj[j] == 0.0 and A.indptr[j] == A.indptr[j + 1]:
                fh.write(f" C{j} COST 0.0\n")
        fh.write("RHS\n")
        if lp.obj0 != 0.0:
            fh.write(f" RHS COST {float(-lp.obj0)!r}\n")
        for i in range(m):
            b = lp.ucon[i] if rtype[i] in ("L", "R", "E") else lp.lcon[i]
            if b != 0.0:
                fh.write(f" RHS R{i} {float(b)!r}\n")
        if "R" in rtype:
            fh.write("RANGES\n")
            for i in range(m):
                if rtype[i] == "R":
                    fh.write(f" RNG R{i} {float(lp.ucon[i] - lp.lcon[i])!r}\n")
        fh.write("BOUNDS\n")
        for j in range(n):
            lo, up = lp.lvar[j], lp.uvar[j]
            if lo == 0.0 and np.isinf(up):
                continue
            if np.isinf(lo) and np.isinf(up):
                fh.write(f" FR BND C{j}\n")
                continue
            if np.isinf(lo):
                fh.write(f" MI BND C{j}\n")
            elif lo != 0.0:
                fh.write(f" LO BND C{j} {float(lo)!r}\n")
            if np.isfinite(up):
                fh.write(f" UP BND C{j} {float(up)!r}\n")
        fh.write("ENDATA\n")
