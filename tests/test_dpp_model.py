"""numpy model of potrf_block_dpp (kernels.hip): the 64 x 64 diagonal block factored panel by panel with lane (g, cc) = column cc
of a 16-column panel and registers = rows, every cross-lane access a `row_newbcast` (lane j of the caller's row of 16 lanes).  The model executes the
kernel's data flow lane by lane -- which register of which lane feeds which update -- so that the two claims the kernel rests on are checked on the CPU:
(1) the multipliers of a step come from ONE triangle of the diagonal block (the entries a lane holds behind the diagonal are never read: they are
perturbed here), (2) the same flow factors a quasi-definite block with the signs known in advance (K2).  The arithmetic of the GPU kernel (v_rcp_f64 +
Newton steps, fused multiply-adds, matrix-core tile updates) is NOT modelled: the GPU suite compares the kernel itself with the older block kernels
(tests/test_gpu_parity.py::test_diagonal_block_kernels_agree)."""
import numpy as np

LANE = np.arange(64)
G, CC = LANE >> 4, LANE & 15


def bcast(reg, j):
    """row_newbcast:j -- every lane reads lane j of its own row of 16 lanes"""
    out = np.empty_like(reg)
    for g in range(4):
        out[16 * g:16 * g + 16] = reg[16 * g + j]
    return out


def factor_block(A, signs, rng):
    n = 64
    As = np.tril(A).copy()
    for p in range(4):                                            # the diagonal 16 x 16 blocks are held full (both triangles)
        s = slice(16 * p, 16 * p + 16); As[s, s] = A[s, s]
    L = np.zeros((n, n))
    for p in range(4):
        nbr = 4 * (3 - p)                                         # registers of rows below the diagonal block per lane
        d = np.array([As[16 * p + k, 16 * p + CC] for k in range(16)])
        b = np.array([As[16 * (p + 1) + G * nbr + k, 16 * p + CC] for k in range(nbr)]).reshape(nbr, 64)
        d = np.where(np.arange(16)[:, None] > CC[None, :], d * (1 + 1e-3 * rng.standard_normal((16, 64))), d)   # behind the diagonal: never read
        pivot = np.zeros(64)
        for j in range(16):
            dj = bcast(d[j], j)
            pivot = np.where(CC == j, dj, pivot)
            w = -((d[j] * (CC > j)) / dj)                         # -A[j][c] / d_j in the lanes of the columns c > j
            for k in range(j + 1, 16):
                d[k] = d[k] + bcast(w, k) * d[j]                  # v_fmac_f64_dpp d[k], w, d[j] row_newbcast:k
            for k in range(nbr):
                b[k] = b[k] + bcast(b[k], j) * w                  # v_fmac_f64_dpp b[k], b[k], w row_newbcast:j
        scale = signs[16 * p + CC] / np.sqrt(np.abs(pivot))       # s_j / sqrt|d_j|, one per lane = column
        assert (signs[16 * p + CC] * pivot > 0).all()
        for j in range(16):
            d[j] = d[j] * bcast(scale, j)                         # lane r, register j <= r: L[r][j]
        b = b * scale
        for ln in range(16):                                      # (row 0 of lanes holds the diagonal block: lane = ROW ln)
            L[16 * p + ln, 16 * p:16 * p + ln + 1] = d[:ln + 1, ln]
        for ln in range(64):
            for k in range(nbr):
                L[16 * (p + 1) + G[ln] * nbr + k, 16 * p + CC[ln]] = b[k, ln]
        S = np.diag(signs[16 * p:16 * p + 16])
        for q in range(p + 1, 4):                                 # trailing 16 x 16 tiles, right-looking (matrix cores in the kernel)
            for g in range(q, 4):
                As[16 * g:16 * g + 16, 16 * q:16 * q + 16] -= L[16 * g:16 * g + 16, 16 * p:16 * p + 16] @ S @ L[16 * q:16 * q + 16, 16 * p:16 * p + 16].T
    return L


def test_positive_definite_block():
    rng = np.random.default_rng(1)
    B = rng.standard_normal((64, 64)); A = B @ B.T + 64 * np.eye(64)
    L = factor_block(A, np.ones(64), rng)
    assert np.abs(L - np.linalg.cholesky(A)).max() <= 1e-13 * np.abs(L).max()
    assert np.abs(np.triu(L, 1)).max() == 0.0


def test_quasi_definite_block_with_large_multipliers():
    """K2: [-D C'; C E] symmetrically permuted, D over 16 orders of magnitude: P K P' = L S L' with S known in advance, multipliers up to 1e4 and more"""
    rng = np.random.default_rng(2)
    m1 = 30
    D = np.diag(10.0 ** rng.uniform(-8, 8, m1)); E = np.diag(10.0 ** rng.uniform(-8, -6, 64 - m1)); C = rng.standard_normal((64 - m1, m1))
    K = np.block([[-D, C.T], [C, E]]); perm = rng.permutation(64); K = K[np.ix_(perm, perm)]
    s = np.sign(np.diag(K))
    L = factor_block(K, s, rng)
    assert np.abs(L @ np.diag(s) @ L.T - K).max() <= 1e-14 * np.abs(K).max()
    assert np.abs(L).max() > 1e3
