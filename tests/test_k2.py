"""K2, the augmented system [-(Theta^-1 + Rp) A'; A Rd] -- the reference's DEFAULT linear system for
Float64 (/root/reference/src/KKT/KKT.jl:134-141, src/KKT/Cholmod/sqd.jl:5-74,
src/KKT/LDLFactorizations/ldlfact.jl:63-139) -- on the device as a signed supernodal Cholesky
P K P' = L S L' (SURVEY.md 8(f)1).  CPU: the analyse phase and the signed schedule through the numpy
emulator against the pinned K2 oracle (oracle/k2_oracle.c).  GPU: the HIP path against that oracle, the
golden vectors, the K1 path and the reference's conformance routine."""
import numpy as np
import pytest
import scipy.sparse as sp

import tulip_jl_amd as tk
from emulate import Emulator
from helpers import SQRT_EPS, golden_tol, ipm_like_data, kkt_residuals, load_golden, random_lp_matrix
from oracle_binding import OracleK1, OracleK2


def k2_setup(A, device=-1, **kw):
    return tk.setup(A, tk.K2(), tk.Backend(device=device, **kw))


def test_k2_analyse_structure():
    A = random_lp_matrix(40, 90, 3, 3)
    kkt = k2_setup(A)
    st = kkt.stats()
    assert (st["m"], st["n"], st["nnzA"]) == (40, 90, A.nnz)
    assert st["nnzS"] == 40 + 90 + A.nnz                       # lower triangle of K: both diagonal blocks + A
    assert tk.linear_system(kkt) == "Augmented system (K2)"
    p = kkt.perm()
    assert sorted(p.tolist()) == list(range(130))
    # nnz(L) of our ordering vs the K2 oracle's own symbolic factorisation with the same permutation
    orc = OracleK2(A, p)
    assert st["nnzL"] >= orc.nnzL                              # relaxed supernodes may add explicit zeros, never lose entries


def test_k2_block_angular_structure_and_emulated_schedule():
    """K2 with a block-angular row partition: the nodes of the augmented system inherit the blocks of the rows, the
    linking constraints and the variables that only touch them form the root front; the signed schedule through the numpy
    emulator against the K2 oracle with the same permutation."""
    from helpers import block_angular
    A, row_block = block_angular(nblocks=5, mk=60, nk=130, m0=12, nnz_in=3, link_prob=0.5, seed=11)
    A = sp.hstack([A, sp.csc_matrix((np.ones(12), (np.arange(A.shape[0] - 12, A.shape[0]), np.arange(12))), shape=(A.shape[0], 12))]).tocsc()
    m, n = A.shape                                          # + 12 columns that touch linking rows only (their slacks)
    kkt = k2_setup(A, row_block=row_block)
    st = kkt.stats()
    assert st["n_blocks"] == 5 and tk.linear_system(kkt) == "Augmented system (K2)"
    p = kkt.perm()
    assert sorted(p.tolist()) == list(range(m + n))
    link_nodes = set((n + np.flatnonzero(row_block < 0)).tolist()) | set(range(n - 12, n))
    assert set(p[-len(link_nodes):].tolist()) == link_nodes    # ordered last: the root front
    th, rp, rd, xp, xd = ipm_like_data(m, n, 5)
    em = Emulator(kkt)
    em.update(th, rp, rd)
    assert em.fail_col is None
    dx, dy = em.solve(xp, xd, A)
    orc = OracleK2(A, p); orc.update(th, rp, rd)
    dxo, dyo = orc.solve(xp, xd)
    assert np.abs(dy - dyo).max() <= 1e-9 * max(1.0, np.abs(dyo).max())
    assert np.abs(dx - dxo).max() <= 1e-9 * max(1.0, np.abs(dxo).max())
    bad = row_block.copy(); bad[0] = 1 - bad[0] if bad[0] in (0, 1) else 0
    with pytest.raises(Exception):                          # a column then couples two diagonal blocks
        k2_setup(A, row_block=bad)


@pytest.mark.parametrize("g", load_golden(), ids=lambda g: g["name"])
def test_k2_emulated_schedule_on_golden_vectors(g):
    kkt = k2_setup(g["A_csc"])
    em = Emulator(kkt)
    em.update(g["theta_inv"], g["regP"], g["regD"])
    assert em.fail_col is None
    dx, dy = em.solve(g["xi_p"], g["xi_d"], g["A_csc"])
    scale = max(np.abs(g["dx"]).max(), np.abs(g["dy"]).max(), 1.0)
    assert np.abs(dx - g["dx"]).max() <= 100 * golden_tol(g) * scale
    assert np.abs(dy - g["dy"]).max() <= 100 * golden_tol(g) * scale


@pytest.mark.parametrize("seed", range(4))
def test_k2_emulated_schedule_vs_k2_oracle(seed):
    A = random_lp_matrix(150 + 200 * seed, 400 + 350 * seed, 3 + seed, 60 + seed, slack=(seed == 2))
    m, n = A.shape
    kkt = k2_setup(A, relax=(seed != 1))
    em = Emulator(kkt)
    th, rp, rd, xp, xd = ipm_like_data(m, n, seed)
    em.update(th, rp, rd)
    assert em.fail_col is None
    dx, dy = em.solve(xp, xd, A)
    orc = OracleK2(A, kkt.perm()); orc.update(th, rp, rd)
    dxo, dyo = orc.solve(xp, xd)
    assert np.abs(dx - dxo).max() <= 1e-9 * max(1.0, np.abs(dxo).max())
    assert np.abs(dy - dyo).max() <= 1e-9 * max(1.0, np.abs(dyo).max())
    r1, r2 = kkt_residuals(A, th, rp, rd, xp, xd, dx, dy)
    assert max(r1, r2) <= 1e-8 * (1 + max(np.abs(xp).max(), np.abs(xd).max()))


def test_k2_wrong_sign_pivot_is_reported():
    """A negative regD makes a constraint pivot negative: not quasi-definite -> reported like spd.jl:47."""
    A = random_lp_matrix(30, 60, 3, 8)
    kkt = k2_setup(A)
    em = Emulator(kkt)
    th, rp, rd, _, _ = ipm_like_data(30, 60, 1)
    rd = rd.copy(); rd[:] = -1e3
    em.update(th, rp, rd)
    assert em.fail_col is not None


# ---------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------
def gpu_compare(A, seed, regime="mid", xtol=1e-9, **kw):
    m, n = A.shape
    kkt = k2_setup(A, device=0, **kw)
    th, rp, rd, xp, xd = ipm_like_data(m, n, seed, regime)
    tk.update(kkt, th, rp, rd)
    dx = np.zeros(n); dy = np.zeros(m)
    tk.solve(dx, dy, kkt, xp, xd)
    orc = OracleK2(A, kkt.perm()); orc.update(th, rp, rd)
    dxo, dyo = orc.solve(xp, xd)
    assert np.abs(dx - dxo).max() <= xtol * max(1.0, np.abs(dxo).max())
    assert np.abs(dy - dyo).max() <= xtol * max(1.0, np.abs(dyo).max())
    return kkt, (th, rp, rd, xp, xd), (dx, dy)


@pytest.mark.gpu
def test_k2_reference_conformance_routine():
    """KKT.run_ls_tests on the reference's K2 fixture (test/KKT/Cholmod/cholmod.jl:8-11)."""
    A = sp.csc_matrix(np.array([[1.0, 0, 1, 0], [0, 1, 0, 1]]))
    kkt = k2_setup(A, device=0)
    r1, r2 = tk.run_ls_tests(A, kkt)
    assert r1 <= SQRT_EPS and r2 <= SQRT_EPS
    assert tk.linear_system(kkt) == "Augmented system (K2)" and tk.backend(kkt) == "HIP (gfx950)"


@pytest.mark.gpu
@pytest.mark.parametrize("g", load_golden(), ids=lambda g: g["name"])
def test_k2_golden_vectors_on_device(g):
    kkt = k2_setup(g["A_csc"], device=0)
    tk.update(kkt, g["theta_inv"], g["regP"], g["regD"])
    dx = np.zeros(g["n"]); dy = np.zeros(g["m"])
    tk.solve(dx, dy, kkt, g["xi_p"], g["xi_d"])
    scale = max(np.abs(g["dx"]).max(), np.abs(g["dy"]).max(), 1.0)
    assert np.abs(dx - g["dx"]).max() <= 100 * golden_tol(g) * scale
    assert np.abs(dy - g["dy"]).max() <= 100 * golden_tol(g) * scale


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_k2_random_sparse_vs_k2_oracle(seed):
    A = random_lp_matrix(300 + 170 * seed, 800 + 300 * seed, 3, 200 + seed, slack=(seed == 1))
    gpu_compare(A, seed, relax=(seed != 2))


@pytest.mark.gpu
def test_k2_block_angular_on_device():
    """K2 on a block-angular LP (two stream groups, root front with both signs) against the K2 oracle and the K1 path."""
    from helpers import block_angular
    A, row_block = block_angular(nblocks=8, mk=300, nk=600, m0=60, nnz_in=3, link_prob=0.5, seed=5)
    kkt, _, _ = gpu_compare(A, 3, row_block=row_block)
    assert kkt.stats()["n_blocks"] == 8
    m, n = A.shape
    th, rp, rd, xp, xd = ipm_like_data(m, n, 3)
    k1 = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=row_block))
    tk.update(k1, th, rp, rd); tk.update(kkt, th, rp, rd)
    d1 = (np.zeros(n), np.zeros(m)); d2 = (np.zeros(n), np.zeros(m))
    tk.solve(d1[0], d1[1], k1, xp, xd); tk.solve(d2[0], d2[1], kkt, xp, xd)
    assert np.abs(d1[1] - d2[1]).max() <= 1e-8 * max(1.0, np.abs(d1[1]).max())
    assert np.abs(d1[0] - d2[0]).max() <= 1e-8 * max(1.0, np.abs(d1[0]).max())


@pytest.mark.gpu
def test_k2_large_fronts_and_k1_agreement():
    """Fronts wider than one block column (wide potrf, single-pass trsm, signed MFMA updates, sweeps), and the
    K1 path on the same data: both solve the same augmented system (systems.jl:34-47)."""
    A = random_lp_matrix(1500, 2500, 6, 11)
    kkt, (th, rp, rd, xp, xd), (dx, dy) = gpu_compare(A, 2, xtol=1e-8)
    assert kkt.symbolic("front_ns").max() > 512
    k1 = tk.setup(A, tk.K1(), tk.Backend(device=0))
    tk.update(k1, th, rp, rd)
    dx1 = np.zeros(2500); dy1 = np.zeros(1500)
    tk.solve(dx1, dy1, k1, xp, xd)
    assert np.abs(dx - dx1).max() <= 1e-8 * max(1.0, np.abs(dx1).max())
    assert np.abs(dy - dy1).max() <= 1e-8 * max(1.0, np.abs(dy1).max())
    # bitwise determinism of the signed path
    dx2 = np.zeros(2500); dy2 = np.zeros(1500)
    tk.update(kkt, th, rp, rd); tk.solve(dx2, dy2, kkt, xp, xd)
    assert (dx2 == dx).all() and (dy2 == dy).all()


@pytest.mark.gpu
def test_k2_free_variables_late_regime_and_failure():
    """theta_inv = 0 exactly on free variables (where K2 is better conditioned than K1), regs = sqrt(eps);
    a wrong-sign pivot is reported and the handle stays usable."""
    A = random_lp_matrix(400, 1200, 3, 9)
    m, n = A.shape
    kkt = k2_setup(A, device=0)
    th, rp, rd, xp, xd = ipm_like_data(m, n, 4, "late")
    tk.update(kkt, th, rp, rd)
    dx = np.zeros(n); dy = np.zeros(m)
    tk.solve(dx, dy, kkt, xp, xd)
    orc = OracleK2(A, kkt.perm()); orc.update(th, rp, rd)
    dxo, dyo = orc.solve(xp, xd)
    r1, r2 = kkt_residuals(A, th, rp, rd, xp, xd, dx, dy)
    o1, o2 = kkt_residuals(A, th, rp, rd, xp, xd, dxo, dyo)
    eps = np.finfo(float).eps
    f1 = 100 * eps * max(float((abs(A) @ np.abs(dx)).max()), float(np.abs(xp).max()))
    f2 = 100 * eps * max(float(((th + rp) * np.abs(dx)).max()), float((abs(A).T @ np.abs(dy)).max()))
    print(f"k2 late: r1 hip={r1:.3e} oracle={o1:.3e} | r2 hip={r2:.3e} oracle={o2:.3e}")
    assert r1 <= 10 * max(o1, f1) and r2 <= 10 * max(o2, f2)
    bad = rd.copy(); bad[:] = -1e3
    with pytest.raises(tk.PosDefException):
        tk.update(kkt, th, rp, bad)
    tk.update(kkt, th, rp, rd)
    dx3 = np.zeros(n); dy3 = np.zeros(m)
    tk.solve(dx3, dy3, kkt, xp, xd)
    assert (dx3 == dx).all() and (dy3 == dy).all()


@pytest.mark.gpu
def test_k2_in_the_ipm_loop():
    """The whole HSD loop on the K2 backend: the reference's default configuration
    (KKT.jl:134-141: K2 for Float64)."""
    import os
    from ipm_harness import HipBackend, OracleBackend, read_free_mps, solve_lp
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    lp = read_free_mps(os.path.join(golden, "stair25.mps"))

    class HipK2(HipBackend):
        def __init__(self, A):
            self.tk = tk
            self.kkt = tk.setup(A, tk.K2(), tk.Backend(device=0))
            self.name = f"{tk.backend(self.kkt)} / {tk.linear_system(self.kkt)}"
    h2, s2 = solve_lp(lp, lambda A: HipK2(A))
    h1, s1 = solve_lp(lp, lambda A: HipBackend(A, device=0))
    from test_lp_configs import STAIR25_OPT
    assert s2["status"] == s1["status"] == "Trm_Optimal"
    assert abs(h2.niter - h1.niter) <= 2
    assert abs(s2["z_primal"] - STAIR25_OPT) <= 1e-6 * (1 + abs(STAIR25_OPT))
    assert max(s2["rho"]) <= SQRT_EPS


@pytest.mark.gpu
@pytest.mark.parametrize("ngpus", [2, 3])
def test_k2_single_process_multi_device_mode(ngpus):
    """The reference's default linear system (K2, KKT.jl:134-141) on a multi-device handle (tlpk_create_multi; the shards share this
    box's single GPU): per-rank ownership of the variable / constraint nodes, replicated root front of both signs, library-owned
    reductions of the root panel and the root right-hand side, results gathered by P2P stores -- against the K2 oracle and against
    the single-device K2 handle."""
    from helpers import block_angular, kkt_residuals
    A, row_block = block_angular(nblocks=7, mk=200, nk=500, m0=50, nnz_in=3, link_prob=0.5, seed=19)
    m, n = A.shape
    th, rp, rd, xp, xd = ipm_like_data(m, n, 19)
    kkt = tk.setup(A, tk.K2(), tk.Backend(device=0, row_block=row_block, ngpus=ngpus, devices=[0] * ngpus))
    assert tk.linear_system(kkt) == "Augmented system (K2)"
    one = tk.setup(A, tk.K2(), tk.Backend(device=0, row_block=row_block))
    orc = OracleK2(A)
    orc.update(th, rp, rd)
    for it in range(2):
        tk.update(kkt, th, rp, rd); tk.update(one, th, rp, rd)
        dx, dy = np.full(n, np.nan), np.full(m, np.nan)
        tk.solve(dx, dy, kkt, xp, xd)
        dx1, dy1 = np.zeros(n), np.zeros(m)
        tk.solve(dx1, dy1, one, xp, xd)
        dxo, dyo = orc.solve(xp, xd)
        sc = max(1.0, np.abs(dxo).max(), np.abs(dyo).max())
        assert np.abs(dx - dxo).max() <= 1e-9 * sc and np.abs(dy - dyo).max() <= 1e-9 * sc
        assert np.abs(dx - dx1).max() <= 1e-10 * sc and np.abs(dy - dy1).max() <= 1e-10 * sc
        r1, r2 = kkt_residuals(A, th, rp, rd, xp, xd, dx, dy)
        assert max(r1, r2) <= 1e-8 * (1 + max(np.abs(xp).max(), np.abs(xd).max()))
        xp = xp[::-1].copy()
    bad = rd.copy(); bad[2] = -1e6                      # a constraint node of the wrong sign in the first shard
    with pytest.raises(tk.PosDefException):
        tk.update(kkt, th, rp, bad)
    tk.update(kkt, th, rp, rd)
    tk.run_ls_tests(A, kkt)


@pytest.mark.gpu
def test_k2_update_skip_lists(monkeypatch):
    """The signed k_update instance with K-segment lists (structural zeros of amalgamated supernodes skipped) against the K2 oracle."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from workloads import block_angular_lp
    monkeypatch.setenv("TLPK_SKIP_MIN_F", "64")
    monkeypatch.setenv("TLPK_SPLITK_TILES", "0")
    A, row_block = block_angular_lp(nblocks=2, mk=2000, nk=4000, m0=150)
    kkt, _, _ = gpu_compare(A, 4, row_block=row_block)
    ut = kkt.symbolic("update_tasks").reshape(-1, 10)
    assert (ut[:, 8] > 0).any(), "no signed update tile skips anything"
