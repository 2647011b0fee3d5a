"""Synthetic LP constraint matrices for bench.py and the tests (BASELINE.json `configs`,
SURVEY.md section 8d).  Pure numpy/scipy, seeded, no reference code involved."""
import numpy as np
import scipy.sparse as sp

SEED = 20260927


def sparse_columns(m, n, nnz_per_col, rng):
    """m x n CSC, ~nnz_per_col N(0,1) entries per column at uniform rows (vectorised; the rare
    duplicate row inside a column is summed, so a column may have one entry fewer)."""
    k = min(nnz_per_col, m)
    rows = rng.integers(0, m, size=(n, k)).ravel()
    cols = np.repeat(np.arange(n), k)
    A = sp.csc_matrix((rng.standard_normal(n * k), (rows, cols)), shape=(m, n))
    A.sum_duplicates()
    A.sort_indices()
    return A


def block_angular_lp(nblocks=64, mk=5000, nk=10000, m0=1000, nnz_in=4, link_prob=0.5, seed=SEED,
                     blocks=None, ineq=False):
    """Config C4: `nblocks` diagonal blocks A_k (mk x nk, nnz_in per column, equality rows) and
    m0 linking rows; each column touches one linking row w.p. link_prob (SURVEY.md 8d proposal,
    seed + k per block).  `blocks` restricts generation to a subset of block ids (same content
    per block as in the full problem).  ineq=True: the block rows are "<=" constraints, i.e. the
    standard-form matrix gets one slack column per block row (A_k = [A0_k I], ipmdata.jl:90-96)
    -- the north-star headline instance is 100 x (2e4 inequality rows x 1e4 vars) + 1e3 linking.
    Returns (A csc, row_block) with linking rows last."""
    ids = list(range(nblocks)) if blocks is None else list(blocks)
    diag, link = [], []
    for k in ids:
        rng = np.random.default_rng(seed + k)
        Ak = sparse_columns(mk, nk, nnz_in, rng)
        cols = np.nonzero(rng.random(nk) < link_prob)[0]
        rows = rng.integers(0, max(m0, 1), size=cols.size)
        vals = rng.standard_normal(cols.size)
        Bk = sp.csc_matrix((vals, (rows, cols)), shape=(m0, nk)) if m0 > 0 else None
        if ineq:
            Ak = sp.hstack([Ak, sp.identity(mk, format="csc")], format="csc")
            if Bk is not None:
                Bk = sp.hstack([Bk, sp.csc_matrix((m0, mk))], format="csc")
        diag.append(Ak); link.append(Bk)
    top = sp.block_diag(diag, format="csc")
    A = sp.vstack([top, sp.hstack(link, format="csc")], format="csc") if m0 > 0 else top
    A.sort_indices()
    row_block = np.concatenate([np.repeat(np.arange(len(ids)), mk), np.full(m0, -1)]).astype(np.int64)
    return A, row_block


def general_sparse_lp(m=50000, n0=None, nnz_col=25, seed=SEED):
    """Config C3 at a feasible scale: A0 in R^{m x n0} (n0 = 0.4 m as in the 5e5 x
    2e5 original), ~nnz_col N(0,1) entries per column at uniform rows, rows are "<=" constraints
    => A = [A0 I].  No block structure: one supernodal tree, ends in a large dense front.  As
    specified (m = 5e5) the factor would be ~0.86 TB; the caller picks m so that it fits."""
    n0 = int(0.4 * m) if n0 is None else n0
    rng = np.random.default_rng(seed)
    A0 = sparse_columns(m, n0, nnz_col, rng)
    A = sp.hstack([A0, sp.identity(m, format="csc")], format="csc")
    A.sort_indices()
    return A


def kernel_inputs(m, n, seed=7, regime="mid"):
    """Kernel-level benchmark inputs: theta_inv = 10^U(-3,3), regP = regD = 1e-4
    ('mid-IPM'), or 10^U(-8,8) with 5 % exact zeros and regs = sqrt(eps) ('late')."""
    rng = np.random.default_rng(seed)
    if regime == "mid":
        th = 10.0 ** rng.uniform(-3, 3, n); rp = np.full(n, 1e-4); rd = np.full(m, 1e-4)
    else:
        th = 10.0 ** rng.uniform(-8, 8, n); th[rng.random(n) < 0.05] = 0.0
        e = float(np.sqrt(np.finfo(float).eps)); rp = np.full(n, e); rd = np.full(m, e)
    return th, rp, rd, rng.standard_normal(m), rng.standard_normal(n)
