"""The dependency-driven form of the blocked factorisation (LK_CHAIN launches, kernels.hip: k_chain; symbolic.cpp: build_chain), on the CPU: the
schedule the library builds is executed by the numpy emulator (tests/emulate.py) -- in ticket order, which asserts that an item only ever waits for
items with SMALLER tickets (the no-deadlock property), and in adversarial orders in which any item whose counters have arrived may run next, which
must give the SAME BITS (the waits alone order every pair of items that touch the same data).  The launch form (TLPK_CHAIN=0) must give those bits
too: the chain runs the same tasks, in the same order per target.  Reference step being scheduled: /root/reference/src/KKT/Cholmod/spd.jl:46
(`cholesky!`), the dense algebra of /root/reference/src/KKT/Dense/lapack.jl:85-95."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from emulate import Emulator  # noqa: E402
import tulip_jl_amd as tk  # noqa: E402
from helpers import block_angular, ipm_like_data  # noqa: E402


def analyse_only(A, **kw):
    return tk.setup(A, tk.K1(), tk.Backend(device=-1, **kw))


def _factor(kkt, seed, rng=None):
    em = Emulator(kkt)
    if rng is not None:
        em.chain_rng = rng
    th, rp, rd, _, _ = ipm_like_data(kkt.m, kkt.n, seed)
    em.update(th, rp, rd)
    assert em.fail_col is None
    return em


def _single_front_lp(m=1500):
    from workloads import general_sparse_lp
    return general_sparse_lp(m)


CASES = {
    # one dense top front of ~1400 columns: macro columns + split-K parts + reductions inside the chain (the pds-class regime)
    "single_front": lambda: (_single_front_lp(1500), None, {"TLPK_MACRO_TILES": "50", "TLPK_CHAIN_MIN_NS": "257"}),
    # a front of eleven block columns whose second macro column is nine block columns wide: macro-column tiles cut by K length (TLPK_KSPLIT_LEN) AND held back
    # until two block columns before their target (build_chain: just-in-time tickets; both knobs are experiments that stay off by default, profiles/r06_chain_variants.txt), the
    # macro column's first block column pulled one block column early
    "macro_jit": lambda: (_single_front_lp(2700), None, {"TLPK_MACRO_TILES": "200", "TLPK_CHAIN_MIN_NS": "257", "TLPK_KSPLIT_LEN": "512", "TLPK_CHAIN_JIT": "1"}),
    # block-angular: the root front (900 columns, no rows below) and the block fronts of a rank that owns few blocks
    "block_angular": lambda: (*block_angular(nblocks=4, mk=300, nk=600, m0=700, nnz_in=3, link_prob=0.9, seed=5), {"TLPK_CHAIN_MIN_NS": "257"}),
    # a front whose last block column is narrow (ns = 2 * 256 + 40) and whose rows below are ragged
    "narrow_last": lambda: (*block_angular(nblocks=2, mk=552, nk=900, m0=90, nnz_in=4, link_prob=0.8, seed=9), {"TLPK_CHAIN_MIN_NS": "257"}),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_chain_schedule_bits_equal_launch_schedule_and_any_order(case, monkeypatch):
    A, rb, env = CASES[case]()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    kkt = analyse_only(A, row_block=rb)
    kinds = kkt.symbolic("factor_launches").reshape(-1, 3)[:, 0].tolist()
    assert 22 in kinds, "no dependency-driven launch on this LP"
    items = kkt.symbolic("chain_items").reshape(-1, 12)
    assert set(items[:, 0].tolist()) >= {0, 1, 2}, "update tiles, diagonal blocks and solve strips are all items"
    assert kkt.symbolic("update_tile64").any(), "the diagonal blocks' short updates run as 64 x 64 tiles"
    st = kkt.stats()
    assert st["chain_launches"] == kinds.count(22) and st["chain_items"] == len(items) and 0 < st["flops_update_alg_chain"] <= st["flops_update_alg"]
    em = _factor(kkt, 3)
    # any order the counters allow
    for seed in (1, 2):
        em2 = _factor(kkt, 3, rng=np.random.default_rng(seed))
        assert np.array_equal(em.Lval, em2.Lval, equal_nan=True), "the result depends on the order in which ready items run"
    # the launch form
    monkeypatch.setenv("TLPK_CHAIN", "0")
    kkt0 = analyse_only(A, row_block=rb)
    assert 22 not in kkt0.symbolic("factor_launches").reshape(-1, 3)[:, 0].tolist()
    em0 = _factor(kkt0, 3)
    assert np.array_equal(em.dense_L(), em0.dense_L()), "chain form and launch form differ"


def test_early_strips_name_the_counter_of_their_diagonal_block(monkeypatch):
    """TLPK_CHAIN_EARLY (default): a strip of a FULL-WIDTH block column does not wait for its diagonal block in the item's wait list; its task names the counter
    that block's role raises on its way (kernels.hip: trsm_task_dma) -- the `sig` of a diagonal-block item with a SMALLER ticket, marked sub = 1, of the same front
    and block column.  Narrow last block columns keep the plain wait.  TLPK_CHAIN_EARLY=0: plain waits everywhere."""
    A, rb, env = CASES["narrow_last"]()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    kkt = analyse_only(A, row_block=rb)
    items = kkt.symbolic("chain_items").reshape(-1, 12)
    trsm = kkt.symbolic("trsm_tasks").reshape(-1, 6)
    potrf = kkt.symbolic("potrf_tasks").reshape(-1, 4)
    early = kkt.symbolic("trsm_early")
    seen_early = seen_plain = 0
    for q, it in enumerate(items):
        role, task, sub, *_w, w2, need2, sig = (int(v) for v in it)
        if role != 2:
            continue
        front, k0, nb = (int(v) for v in trsm[task][:3])
        if nb == 256:
            assert early[task] > 0 and w2 == -1, "a full-width strip enters early"
            owner = [p for p in range(q) if int(items[p][0]) == 1 and int(items[p][11]) == early[task] - 1]
            assert len(owner) == 1, "the counter belongs to ONE diagonal-block item with a smaller ticket"
            pt = potrf[int(items[owner[0]][1])]
            assert int(items[owner[0]][2]) == 1 and (int(pt[0]), int(pt[1]), int(pt[2])) == (front, k0, 256)
            seen_early += 1
        else:
            assert early[task] == 0 and w2 >= 0 and need2 == 1
            seen_plain += 1
    assert seen_early and seen_plain
    monkeypatch.setenv("TLPK_CHAIN_EARLY", "0")
    kkt0 = analyse_only(A, row_block=rb)
    it0 = kkt0.symbolic("chain_items").reshape(-1, 12)
    assert not kkt0.symbolic("trsm_early").any() and (it0[it0[:, 0] == 2][:, 9] >= 0).all() and not it0[it0[:, 0] == 1][:, 2].any()
    # same tasks, same tickets: only the waits differ
    assert np.array_equal(np.delete(items, [2, 9, 10], axis=1), np.delete(it0, [2, 9, 10], axis=1))


def test_chain_only_where_it_was_measured_to_help(monkeypatch):
    """Auto rule: levels with at most 16 of the rank's fronts wider than one block column; TLPK_CHAIN=1 forces every such level, 0 none; the older diagonal-block
    kernels (TLPK_POTRF_MODE != 3) keep the launches."""
    A, rb = block_angular(nblocks=20, mk=300, nk=600, m0=40, nnz_in=3, link_prob=0.5, seed=2)
    kinds = lambda k: k.symbolic("factor_launches").reshape(-1, 3)[:, 0].tolist()  # noqa: E731
    k_auto = analyse_only(A, row_block=rb)
    ns = k_auto.symbolic("front_ns")
    if (ns > 256).sum() > 16:
        assert 22 not in kinds(k_auto)
    monkeypatch.setenv("TLPK_CHAIN", "1")
    k_on = analyse_only(A, row_block=rb)
    assert (22 in kinds(k_on)) == bool((ns > 256).any())
    monkeypatch.setenv("TLPK_POTRF_MODE", "0")
    assert 22 not in kinds(analyse_only(A, row_block=rb))


# ---- on the device ---------------------------------------------------------------------------------------------------------------------
def _gpu_factor(A, rb, system, seed, repeats=1):
    import tulip_jl_amd as tk
    kkt = tk.setup(A, tk.K2() if system == "K2" else tk.K1(), tk.Backend(device=0, row_block=rb))
    th, rp, rd, xp, xd = ipm_like_data(kkt.m, kkt.n, seed)
    outs = []
    for _ in range(repeats):
        tk.update(kkt, th, rp, rd)
        dx = np.zeros(kkt.n); dy = np.zeros(kkt.m)
        tk.solve(dx, dy, kkt, xp, xd)
        outs.append((kkt.factor_panels().copy(), dx, dy))
    kinds = kkt.symbolic("factor_launches").reshape(-1, 3)[:, 0].tolist()
    st = kkt.stats()
    kkt.close()
    return outs, kinds, st


@pytest.mark.gpu
@pytest.mark.parametrize("system", ["K1", "K2"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_chain_on_device_bits_equal_launch_form_run_after_run(case, system, monkeypatch):
    """k_chain against the launches it replaces, on the MI355X: the factor (every stored entry) and the solution must be BIT-identical to the launch form
    (same tiles, same K ranges, same order of the adders of every target), and identical run after run (5 factorisations on one handle: whatever
    order the workgroups draw and finish their items in, the completion counters fix the order of everything that touches the same data).  A stale read
    across workgroups -- a missing release / acquire -- shows up here as a differing bit."""
    A, rb, env = CASES[case]()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    outs, kinds, st = _gpu_factor(A, rb, system, 3, repeats=5)
    assert 22 in kinds and st["chain_items"] > 0
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0], equal_nan=True) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])
    monkeypatch.setenv("TLPK_CHAIN", "0")
    outs0, kinds0, _ = _gpu_factor(A, rb, system, 3)
    assert 22 not in kinds0
    assert np.array_equal(outs0[0][0], outs[0][0], equal_nan=True), "chain form and launch form differ on the device"
    assert np.array_equal(outs0[0][1], outs[0][1]) and np.array_equal(outs0[0][2], outs[0][2])


@pytest.mark.gpu
def test_chain_under_serialised_launches_and_small_grids(monkeypatch):
    """One launch with in-kernel dependencies must not care how many workgroups are resident: 1, 3 and 40 workgroups (TLPK_CHAIN_GRID) walk the same
    tickets to the same bits -- with ONE workgroup every wait is already satisfied when its item is drawn (the ticket-order property, on the device) --
    and so does the single-stream mode (TLPK_SERIAL=1: what a counter-collecting profiler sees)."""
    A, rb, env = CASES["single_front"]()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ref, kinds, _ = _gpu_factor(A, rb, "K1", 7)
    assert 22 in kinds
    for grid in ("1", "3", "40"):
        monkeypatch.setenv("TLPK_CHAIN_GRID", grid)
        o, _, _ = _gpu_factor(A, rb, "K1", 7)
        assert np.array_equal(o[0][0], ref[0][0], equal_nan=True), grid
    monkeypatch.delenv("TLPK_CHAIN_GRID")
    monkeypatch.setenv("TLPK_SERIAL", "1")
    o, _, _ = _gpu_factor(A, rb, "K1", 7)
    assert np.array_equal(o[0][0], ref[0][0], equal_nan=True)


@pytest.mark.gpu
def test_strip_variants_of_the_chain_same_bits(monkeypatch):
    """The strips of the dependency-driven launches: ring of four operand images + early entry (default), plain waits (TLPK_CHAIN_EARLY=0), ring of two images
    (no dynamic LDS: two workgroups per CU) with and without early entry -- one arithmetic, one result."""
    A, rb, env = CASES["single_front"]()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ref, kinds, _ = _gpu_factor(A, rb, "K1", 11)
    assert 22 in kinds
    for var in ({"TLPK_CHAIN_EARLY": "0"}, {"TLPK_CHAIN_DYNLDS": "0", "TLPK_CHAIN_GRID": "512"}, {"TLPK_CHAIN_DYNLDS": "0", "TLPK_CHAIN_GRID": "512", "TLPK_CHAIN_EARLY": "0"}):
        for k, v in var.items():
            monkeypatch.setenv(k, v)
        o, _, _ = _gpu_factor(A, rb, "K1", 11, repeats=2)
        for r in o:
            assert np.array_equal(r[0], ref[0][0], equal_nan=True) and np.array_equal(r[1], ref[0][1]) and np.array_equal(r[2], ref[0][2]), var
        for k in var:
            monkeypatch.delenv(k)


@pytest.mark.gpu
def test_an_update_that_gave_up_is_replayed_and_leaves_no_trace(monkeypatch):
    """A dependency-driven launch that gives up waiting ends the update with TLPK_INTERNAL (bounded spins: nothing hangs).  TLPK_CHAIN_FAULT=k starts update k of a handle
    with the give-up flag set, as if a workgroup had just given up.  Without the replay (TLPK_CHAIN_RETRY=0) that update fails and the NEXT one must work and give the bits
    of a healthy update (the flag used to stay set on the device: every later update of the handle failed within milliseconds); with it (default) tlpk_update replays the
    update once and the caller sees nothing but `chain_retries` = 1."""
    import tulip_jl_amd as tk
    A, rb, env = CASES["single_front"]()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ref, kinds, _ = _gpu_factor(A, rb, "K1", 5)
    assert 22 in kinds
    monkeypatch.setenv("TLPK_CHAIN_FAULT", "2")
    monkeypatch.setenv("TLPK_GRAPH", "0")          # (the injection sits in the enqueue path: a replayed graph would not pass through it)
    for retry in ("0", "1"):
        monkeypatch.setenv("TLPK_CHAIN_RETRY", retry)
        kkt = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb))
        th, rp, rd, xp, xd = ipm_like_data(kkt.m, kkt.n, 5)
        failed = []
        for it in range(5):
            try:
                tk.update(kkt, th, rp, rd)
            except RuntimeError as e:
                assert "gave up waiting" in str(e)
                failed.append(it)
                continue
            assert np.array_equal(kkt.factor_panels(), ref[0][0], equal_nan=True), (retry, it)
        retries = int(kkt.symbolic("chain_retries")[0])
        assert (failed, retries) == (([2], 0) if retry == "0" else ([], 1)), (retry, failed, retries)
        kkt.close()
