"""Pin the CPU oracle (oracle/k1_oracle.c) before anything is compared against it:
reference fixture, golden vectors, dense numpy solves, structural checks of S and L."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import SQRT_EPS, golden_tol, ipm_like_data, kkt_residuals, load_golden, random_lp_matrix
from oracle_binding import OracleK1, OraclePosDefError


def test_reference_fixture_run_ls_tests():
    """/root/reference/test/KKT/Cholmod/cholmod.jl:3-16 through src/KKT/Test/test.jl:26-43."""
    A = sp.csc_matrix(np.array([[1.0, 0, 1, 0], [0, 1, 0, 1]]))
    o = OracleK1(A)
    th = np.ones(4); rp = np.ones(4); rd = np.ones(2)
    o.update(th, rp, rd)
    dx, dy = o.solve(np.ones(2), np.ones(4))
    rp_, rd_ = kkt_residuals(A, th, rp, rd, np.ones(2), np.ones(4), dx, dy)
    assert rp_ <= SQRT_EPS and rd_ <= SQRT_EPS          # test.jl:41-42, atol = sqrt(eps)
    np.testing.assert_allclose(dy, [1.0, 1.0], rtol=0, atol=1e-15)
    np.testing.assert_allclose(dx, np.zeros(4), rtol=0, atol=1e-15)


@pytest.mark.parametrize("g", load_golden(), ids=lambda g: g["name"])
def test_golden_vectors(g):
    o = OracleK1(g["A_csc"])
    o.update(g["theta_inv"], g["regP"], g["regD"])
    dx, dy = o.solve(g["xi_p"], g["xi_d"])
    scale = max(np.abs(g["dx"]).max(), np.abs(g["dy"]).max(), 1.0)
    # Tolerance relative to the solution size.  The K1 reduction solves with S, so the
    # attainable forward accuracy of ANY backward-stable Cholesky (the reference's CHOLMOD path
    # included -- spd.jl has no refinement, spd.jl:68) is ~eps*cond(S); cond(S) is stored in the
    # fixture.  Gate: 10*eps*cond(S), floor 1e-13.
    tol = max(1e-13, 10 * np.finfo(float).eps * g["cond_S"])
    assert np.abs(dx - g["dx"]).max() <= tol * scale
    assert np.abs(dy - g["dy"]).max() <= tol * scale
    if "cholS" in g:                                     # KAT-2: S and chol(S) themselves
        S = o.get_S().toarray()
        np.testing.assert_allclose(S, np.tril(np.array(g["S"])), rtol=1e-15, atol=0)
        np.testing.assert_allclose(o.get_L().toarray(), np.array(g["cholS"]), rtol=1e-15, atol=0)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("use_perm", [False, True])
def test_random_vs_dense(seed, use_perm):
    m, n = 40 + 7 * seed, 90 + 5 * seed
    A = random_lp_matrix(m, n, 3, seed)
    th, rp, rd, xp, xd = ipm_like_data(m, n, seed)
    perm = np.random.default_rng(seed).permutation(m) if use_perm else None
    o = OracleK1(A, perm)
    o.update(th, rp, rd)
    dx, dy = o.solve(xp, xd)
    Ad = A.toarray()
    D = 1.0 / (th + rp)
    S = Ad @ np.diag(D) @ Ad.T + np.diag(rd)
    dy_ref = np.linalg.solve(S, xp + Ad @ (D * xd))
    dx_ref = D * (Ad.T @ dy_ref - xd)
    np.testing.assert_allclose(dy, dy_ref, rtol=1e-9, atol=1e-11 * np.abs(dy_ref).max())
    np.testing.assert_allclose(dx, dx_ref, rtol=1e-9, atol=1e-11 * np.abs(dx_ref).max())
    # S on the fixed pattern and L*L' = P S P'
    P = np.arange(m) if perm is None else perm
    Sp = S[np.ix_(P, P)]
    np.testing.assert_allclose(o.get_S().toarray(), np.tril(Sp), rtol=1e-12, atol=1e-13)
    L = o.get_L().toarray()
    np.testing.assert_allclose(L @ L.T, Sp, rtol=1e-10, atol=1e-10 * np.abs(Sp).max())
    assert o.nnzL >= o.nnzS


def test_free_variables_and_stored_copies():
    """theta_inv = 0 for free variables (HSD/step.jl:24-26) => D_j = 1/regP_j; update! copies
    its inputs (spd.jl:36-38) so later mutation by the caller must not matter."""
    m, n = 12, 30
    A = random_lp_matrix(m, n, 3, 5)
    th, rp, rd, xp, xd = ipm_like_data(m, n, 5, "late")
    th[:5] = 0.0
    o = OracleK1(A)
    th2, rp2, rd2 = th.copy(), rp.copy(), rd.copy()
    o.update(th2, rp2, rd2)
    th2[:] = 7.0; rp2[:] = 3.0; rd2[:] = 9.0
    dx, dy = o.solve(xp, xd)
    r1, r2 = kkt_residuals(A, th, rp, rd, xp, xd, dx, dy)
    assert max(r1, r2) <= 1e-8 * (1 + max(np.abs(xp).max(), np.abs(xd).max()) * 1e8)


def test_not_posdef_is_reported_and_object_reusable():
    """spd.jl:46-47 -> PosDefException; the IPM retries with larger regularisation
    (HSD/step.jl:35-49) so the object must stay usable."""
    A = sp.csc_matrix(np.array([[1.0, 2.0], [2.0, 4.0]]))          # rank-1 A A'
    o = OracleK1(A)
    with pytest.raises(OraclePosDefError):
        o.update(np.ones(2), np.zeros(2), np.array([0.0, -1.0]))
    o.update(np.ones(2), np.zeros(2), np.array([1e-3, 1e-3]))
    dx, dy = o.solve(np.ones(2), np.ones(2))
    assert np.isfinite(dx).all() and np.isfinite(dy).all()


def test_empty_and_degenerate_shapes():
    # a row of A that is structurally empty: S_ii = regD_i only
    A = sp.csc_matrix(np.array([[1.0, 0.0, 2.0], [0.0, 0.0, 0.0]]))
    o = OracleK1(A)
    o.update(np.ones(3), np.ones(3), np.array([0.5, 0.25]))
    dx, dy = o.solve(np.array([1.0, 1.0]), np.zeros(3))
    assert dy[1] == pytest.approx(4.0)
    # n = 0 columns
    A0 = sp.csc_matrix((3, 0))
    o = OracleK1(A0)
    o.update(np.ones(0), np.ones(0), np.full(3, 2.0))
    dx, dy = o.solve(np.ones(3), np.ones(0))
    np.testing.assert_allclose(dy, 0.5)
    assert dx.shape == (0,)


# ------------------------------------------------------------------------------------------------
# K2 (augmented system) oracle, oracle/k2_oracle.c: pinned ahead of the device path it will check
# (SURVEY.md section 8(f)1).  K1 and K2 solve the SAME augmented system (KKT.jl:70-75), so the golden
# dx, dy apply to both, and the two oracles must agree with each other.
# ------------------------------------------------------------------------------------------------
def test_k2_reference_fixture_run_ls_tests():
    """/root/reference/test/KKT/Cholmod/cholmod.jl:3-16: run_ls_tests on the K2 solver, same 2 x 4 fixture."""
    from oracle_binding import OracleK2
    A = sp.csc_matrix(np.array([[1.0, 0, 1, 0], [0, 1, 0, 1]]))
    o = OracleK2(A)
    th = np.ones(4); rp = np.ones(4); rd = np.ones(2)
    o.update(th, rp, rd)
    dx, dy = o.solve(np.ones(2), np.ones(4))
    rp_, rd_ = kkt_residuals(A, th, rp, rd, np.ones(2), np.ones(4), dx, dy)
    assert rp_ <= SQRT_EPS and rd_ <= SQRT_EPS
    np.testing.assert_allclose(dy, [1.0, 1.0], rtol=0, atol=1e-15)
    np.testing.assert_allclose(dx, np.zeros(4), rtol=0, atol=1e-15)
    d = o.D()
    assert (d[:4] < 0).all() and (d[4:] > 0).all()       # quasi-definite: n negative, m positive pivots


@pytest.mark.parametrize("g", load_golden(), ids=lambda g: g["name"])
def test_k2_golden_vectors(g):
    from oracle_binding import OracleK2
    o = OracleK2(g["A_csc"])
    o.update(g["theta_inv"], g["regP"], g["regD"])
    dx, dy = o.solve(g["xi_p"], g["xi_d"])
    scale = max(np.abs(g["dx"]).max(), np.abs(g["dy"]).max(), 1.0)
    tol = max(1e-13, 10 * np.finfo(float).eps * g["cond_S"])
    assert np.abs(dx - g["dx"]).max() <= tol * scale
    assert np.abs(dy - g["dy"]).max() <= tol * scale


@pytest.mark.parametrize("seed", range(3))
def test_k2_agrees_with_k1_and_dense(seed):
    from oracle_binding import OracleK2
    m, n = 60 + 25 * seed, 150 + 40 * seed
    A = random_lp_matrix(m, n, 3, 70 + seed)
    th, rp, rd, xp, xd = ipm_like_data(m, n, seed)
    rng = np.random.default_rng(seed)
    perm = rng.permutation(m + n)                         # any symmetric permutation works for SQD matrices
    for pp in (None, perm):
        o2 = OracleK2(A, pp)
        o2.update(th, rp, rd)
        dx2, dy2 = o2.solve(xp, xd)
        K = np.block([[-np.diag(th + rp), A.toarray().T], [A.toarray(), np.diag(rd)]])
        ref = np.linalg.solve(K, np.concatenate([xd, xp]))
        sc = max(1.0, np.abs(ref).max())
        assert np.abs(dx2 - ref[:n]).max() <= 1e-9 * sc and np.abs(dy2 - ref[n:]).max() <= 1e-9 * sc
        d = o2.D()
        neg = np.array([(pp[k] if pp is not None else k) < n for k in range(m + n)])
        assert (d[neg] < 0).all() and (d[~neg] > 0).all()
    o1 = OracleK1(A); o1.update(th, rp, rd)
    dx1, dy1 = o1.solve(xp, xd)
    assert np.abs(dx1 - dx2).max() <= 1e-8 * max(1.0, np.abs(dx1).max())
    assert np.abs(dy1 - dy2).max() <= 1e-8 * max(1.0, np.abs(dy1).max())


def test_k2_zero_pivot_and_reuse():
    """A failed factorisation reports the pivot and leaves the object reusable (HSD/step.jl:35-49)."""
    from oracle_binding import OracleK2, OracleZeroPivotError
    A = random_lp_matrix(20, 45, 3, 5)
    th, rp, rd, xp, xd = ipm_like_data(20, 45, 1)
    o = OracleK2(A)
    bad = th.copy(); bad[3] = -rp[3]                       # -(theta + regP) = 0 on variable 3
    with pytest.raises(OracleZeroPivotError):
        o.update(bad, rp, rd)
    o.update(th, rp, rd)
    dx, dy = o.solve(xp, xd)
    r1, r2 = kkt_residuals(A, th, rp, rd, xp, xd, dx, dy)
    assert r1 <= 1e-9 * (1 + np.abs(xp).max()) * max(1, np.abs(dy).max()) and r2 <= 1e-9 * (1 + np.abs(xd).max()) * max(1, np.abs(dx).max())


def test_k2_signed_cholesky_design():
    """Executable design note for the device K2 path: a symmetric quasi-definite
    matrix K factors as L*S*L' with S = diag(+-1) known from the node type, and the unit-lower
    elimination steps of the Cholesky kernels (a_rc -= a_rj*a_cj/d) are sign-agnostic: only the column
    scaling uses |d| and sign(d).  The blocked form used on the device follows: trailing updates are
    T -= X*S_K*X', panel solves X = B*L11^-T*S11, and the solve is x = L^-T S L^-1 b."""
    rng = np.random.default_rng(3)
    m, n = 7, 11
    A = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.5)
    K = np.block([[-np.diag(rng.uniform(0.5, 2.0, n)), A.T], [A, np.diag(rng.uniform(0.1, 1.0, m))]])
    N = n + m
    perm = rng.permutation(N)
    Kp = K[np.ix_(perm, perm)]
    s_expected = np.where(perm < n, -1.0, 1.0)
    # column-by-column, exactly the arithmetic of potrf_block with signed pivots
    a = np.tril(Kp).copy()
    L = np.zeros((N, N)); s = np.zeros(N)
    for j in range(N):
        d = a[j, j]
        s[j] = np.sign(d)
        assert s[j] == s_expected[j]                       # quasi-definite: the sign is known a priori
        mult = a[j + 1:, j] / d                            # unit-lower multipliers, sign-agnostic
        a[j + 1:, j + 1:] -= np.tril(np.outer(mult, a[j + 1:, j]))
        L[j, j] = np.sqrt(abs(d))
        L[j + 1:, j] = a[j + 1:, j] * s[j] / L[j, j]
    np.testing.assert_allclose(L @ np.diag(s) @ L.T, Kp, atol=1e-12)
    # blocked (supernodal) form: two block columns
    k = 8
    L11 = L[:k, :k]; S1 = np.diag(s[:k])
    X = Kp[k:, :k] @ np.linalg.inv(L11).T @ S1              # X = B * L11^-T * S11   (S^-1 = S)
    np.testing.assert_allclose(X, L[k:, :k], atol=1e-12)
    T = Kp[k:, k:] - X @ S1 @ X.T                           # trailing update with the signs of the K columns
    np.testing.assert_allclose(np.tril(T), np.tril(L[k:, k:] @ np.diag(s[k:]) @ L[k:, k:].T), atol=1e-12)
    # solve
    b = rng.standard_normal(N)
    y = np.linalg.solve(L, b)
    x = np.linalg.solve(L.T, s * y)
    np.testing.assert_allclose(Kp @ x, b, atol=1e-10)


# ---------------------------------------------------------------------------------------------
# CPU supernodal comparator (oracle/k1_supernodal.c): pinned against the simplicial oracle, the
# reference fixture and the golden vectors before bench.py may use it as cpu_baseline
# ---------------------------------------------------------------------------------------------
def _analyse_only(A, **kw):
    import tulip_jl_amd as tk
    return tk.setup(A, tk.K1(), tk.Backend(device=-1, **kw))


def test_supernodal_reference_fixture():
    from oracle_binding import SupernodalK1
    A = sp.csc_matrix(np.array([[1.0, 0, 1, 0], [0, 1, 0, 1]]))
    sn = SupernodalK1(A, _analyse_only(A))
    sn.update(np.ones(4), np.ones(4), np.ones(2))
    dx, dy = sn.solve(np.ones(2), np.ones(4))
    np.testing.assert_allclose(dy, [1.0, 1.0], atol=1e-15)
    np.testing.assert_allclose(dx, 0.0, atol=1e-15)


@pytest.mark.parametrize("g", load_golden(), ids=lambda g: g["name"])
def test_supernodal_golden_vectors(g):
    from oracle_binding import SupernodalK1
    sn = SupernodalK1(g["A_csc"], _analyse_only(g["A_csc"]))
    sn.update(g["theta_inv"], g["regP"], g["regD"])
    dx, dy = sn.solve(g["xi_p"], g["xi_d"])
    scale = max(np.abs(g["dx"]).max(), np.abs(g["dy"]).max(), 1.0)
    assert np.abs(dx - g["dx"]).max() <= golden_tol(g) * scale
    assert np.abs(dy - g["dy"]).max() <= golden_tol(g) * scale


@pytest.mark.parametrize("threads", [1, 0])
@pytest.mark.parametrize("seed", range(3))
def test_supernodal_matches_simplicial_oracle(seed, threads):
    """L entry by entry, dx, dy: two independent CPU factorisations (simplicial left-looking C vs
    multifrontal on LAPACK/BLAS) of the same permuted matrix; block-angular structure on seed 2."""
    from emulate import panels_to_dense_L
    from helpers import block_angular
    from oracle_binding import OracleK1, SupernodalK1
    if seed == 2:
        A, rb = block_angular(nblocks=5, mk=120, nk=260, m0=30, nnz_in=3, link_prob=0.5, seed=3)
        kk = _analyse_only(A, row_block=rb)
    else:
        A = random_lp_matrix(250 + 400 * seed, 700 + 600 * seed, 3 + 2 * seed, 40 + seed, slack=(seed == 1))
        kk = _analyse_only(A)
    m, n = A.shape
    th, rp, rd, xp, xd = ipm_like_data(m, n, seed)
    orc = OracleK1(A, kk.perm()); orc.update(th, rp, rd)
    dxo, dyo = orc.solve(xp, xd)
    sn = SupernodalK1(A, kk, threads=threads)
    sn.update(th, rp, rd)
    dx, dy = sn.solve(xp, xd)
    L = panels_to_dense_L(kk, sn.factor_panels()); Lo = orc.get_L().toarray()
    assert np.abs(L - Lo).max() <= 1e-12 * np.abs(Lo).max()
    assert np.abs(dy - dyo).max() <= 1e-10 * max(1.0, np.abs(dyo).max())
    assert np.abs(dx - dxo).max() <= 1e-10 * max(1.0, np.abs(dxo).max())


def test_supernodal_not_posdef_and_reuse():
    from oracle_binding import OraclePosDefError, SupernodalK1
    A = random_lp_matrix(50, 120, 3, 2)
    sn = SupernodalK1(A, _analyse_only(A))
    th, rp, rd, xp, xd = ipm_like_data(50, 120, 0)
    bad = rd.copy(); bad[7] = -1e6
    with pytest.raises(OraclePosDefError):
        sn.update(th, rp, bad)
    with pytest.raises(RuntimeError):
        sn.solve(xp, xd)
    sn.update(th, rp, rd)
    dx, dy = sn.solve(xp, xd)
    r1, r2 = kkt_residuals(A, th, rp, rd, xp, xd, dx, dy)
    assert max(r1, r2) <= 1e-8 * (1 + max(np.abs(xp).max(), np.abs(xd).max()))
