"""CPU emulation (numpy) of the device schedule of libtlpk, driven ONLY by the symbolic arrays
and task lists the library exports (tlpk_symbolic_get).  It executes exactly what the HIP
kernels execute -- assembly lists, extend-add, potrf/trsm/update tasks, forward/backward solve
tasks, in launch order -- so the host analyse phase (ordering, supernodes, relative indices,
schedules) can be validated in the CPU-only test run.  Test infrastructure: never imported by
the product."""
import numpy as np

LK = dict(EXTEND_ADD=0, POTRF=1, TRSM=2, UPDATE=3, FWD_GATHER=4, FWD_DIAG=5, FWD_UPDATE=6,
          BWD_UPDATE=7, BWD_DIAG=8, ALLREDUCE=9, POTRF_WIDE=10, SIDE_FORK=11, SIDE_JOIN=12,
          UPDATE_REDUCE=13, TRSM_THIN=14, FWD_SMALL=15, BWD_SMALL=16, POTRF_SMALL=17, FWD_SWEEP=18, BWD_SWEEP=19, FRONT_ASSEMBLE=20, WAIT_UPPER=21, CHAIN=22, UPDATE_T64=23)


class Emulator:
    def __init__(self, kkt):
        g = kkt.symbolic
        self.kkt = kkt
        self.m, self.n = kkt.m, kkt.n
        self.perm = g("perm")
        # K2 (augmented system): the factored matrix has order n + m; S = diag(sign) with -1 on variable nodes
        self.k2 = getattr(kkt, "system", 0) == 1
        self.sign = np.where(self.perm < self.n, -1.0, 1.0) if self.k2 else None
        self.f = g("front_f"); self.ns = g("front_ns"); self.col0 = g("front_col0"); self.lda = g("front_lda")
        self.parent = g("front_parent"); self.loff = g("front_loff"); self.rowoff = g("front_rowoff")
        self.reloff = g("front_reloff"); self.child_ptr = g("front_child_ptr"); self.nchild = g("front_nchild")
        self.local = g("front_local")
        self.single = g("front_single")
        self.front_fa = g("front_fa")          # panel formed by k_front_assemble: no zero-fill, not in k_assemble, no panel-part extend-add
        self.front_upper = g("front_upper")    # zero-fill + assembly deferred to a stream of their own: nothing may touch the panel before the group's WAIT_UPPER marker
        self.front_group = g("front_group")
        Sp = g("s_colptr")
        self.col_of_entry = np.repeat(np.arange(len(Sp) - 1), np.diff(Sp))            # permuted column of every entry of S
        fa_col = np.zeros(len(Sp) - 1, dtype=bool)
        for s_ in np.nonzero(self.front_fa)[0]:
            fa_col[self.col0[s_]: self.col0[s_] + self.ns[s_]] = True
        self.fa_entry = fa_col[self.col_of_entry]
        self.row_local = g("row_local"); self.col_local = g("col_local")
        self.root_front = int(g("root_front")[0])
        self.rank = kkt.backend_options.rank
        self.rowidx = g("rowidx"); self.rel = g("rel"); self.children = g("children")
        self.ea_tab = g("ea_tab"); self.eatab = g("front_eatab")
        self.ucoff = g("front_ucoff"); self.gth_ptr = g("gth_ptr"); self.gth_src = g("gth_src")
        self.s_target = g("s_target"); self.s_diag_row = g("s_diag_row")
        self.pair_ptr = g("pair_ptr"); self.pair_j = g("pair_j")
        from tulip_jl_amd import _lib
        self.pair_w = _lib.symbolic_array_f64(kkt._h, "pair_w")
        self.trsm_early = g("trsm_early")
        self.tasks = {
            LK["EXTEND_ADD"]: g("ea_tasks").reshape(-1, 6),
            LK["FRONT_ASSEMBLE"]: g("fa_tasks").reshape(-1, 4),
            LK["POTRF"]: g("potrf_tasks").reshape(-1, 4),
            LK["TRSM"]: g("trsm_tasks").reshape(-1, 6),      # (early entry of a strip: self.trsm_early)
            LK["UPDATE"]: np.concatenate([g("update_tasks").reshape(-1, 10), g("update_tile64").reshape(-1, 1)], axis=1),      # + 1 = a 64 x 64 tile
            LK["UPDATE_REDUCE"]: g("reduce_tasks").reshape(-1, 8),
            LK["FWD_GATHER"]: g("fwd_gather_tasks").reshape(-1, 6),
            LK["FWD_DIAG"]: g("fwd_diag_tasks").reshape(-1, 6),
            LK["FWD_UPDATE"]: g("fwd_update_tasks").reshape(-1, 6),
            LK["BWD_UPDATE"]: g("bwd_update_tasks").reshape(-1, 6),
            LK["FWD_SMALL"]: g("fwd_small_tasks").reshape(-1, 6),
            LK["BWD_SMALL"]: g("bwd_small_tasks").reshape(-1, 6),
            LK["FWD_SWEEP"]: g("fwd_sweep_tasks").reshape(-1, 6),
            LK["BWD_SWEEP"]: g("bwd_sweep_tasks").reshape(-1, 6),
        }
        self.chain_items = g("chain_items").reshape(-1, 12)      # items of the dependency-driven launches (LK_CHAIN): role, task, sub, three waits, signal
        self.chain_cnt = np.zeros(int(g("chain_counters")[0]) + 1, dtype=np.int64)
        self.upd_seg = g("upd_seg")     # K-segment lists of the update tasks that skip structurally zero slabs
        self.flagoff = g("front_flagoff")
        self.factor_launches = g("factor_launches").reshape(-1, 3)
        self.fwd_launches = g("fwd_launches").reshape(-1, 3)
        self.bwd_launches = g("bwd_launches").reshape(-1, 3)
        st = kkt.stats()
        self.lval_len = st["nnzL_stored"]
        self.Lval = np.zeros(max(self.lval_len, 1))
        self.U = {}          # front -> rs x rs array (lower part meaningful)
        self.spart = {}      # split-K scratch slot -> partial tile
        self.fail_col = None

    # The library stores a panel by 64-column slices (tlpk_host.hpp: pk_off): slice b keeps the rows from 64 b down, with leading
    # dimension lda - 64 b.  The assembly targets (s_target) address that packed storage; the arithmetic below works on an unpacked
    # f x ns copy per front, made from the packed storage at the first access after an assembly (blocks above the diagonal blocks: 0).
    def panel(self, s):
        cache = self.__dict__.setdefault("_P", {})
        if s not in cache:
            cache[s] = unpack_panel(self.Lval, int(self.loff[s]), int(self.f[s]), int(self.ns[s]), int(self.lda[s]))
        return cache[s]

    def rows(self, s):
        return self.rowidx[self.rowoff[s]: self.rowoff[s] + self.f[s]]

    def relidx(self, s):
        rs = self.f[s] - self.ns[s]
        return self.rel[self.reloff[s]: self.reloff[s] + rs]

    def kids(self, s):
        return self.children[self.child_ptr[s]: self.child_ptr[s] + self.nchild[s]]

    def front_get(self, s, r, c):
        ns = self.ns[s]
        return self.panel(s)[r, c] if c < ns else self.U[s][r - ns, c - ns]

    # ---- update! ----
    def update(self, theta, regP, regD, stop_at_marker=False):
        self.D = np.concatenate([theta + regP, [1.0]]) if self.k2 else 1.0 / (theta + regP)   # K2: D2 = [theta + regP ; 1]
        self.regD = np.asarray(regD, dtype=float)
        self.Lval[:] = 0.0
        self._P = {}
        contrib = self.pair_w * self.D[self.pair_j]
        nent = len(self.s_target)
        vals = np.add.reduceat(np.concatenate([contrib, [0.0]]), np.minimum(self.pair_ptr[:-1], len(contrib)))
        vals[self.pair_ptr[:-1] == self.pair_ptr[1:]] = 0.0
        self._svals = {}
        for e in range(nent):
            if self.s_target[e] < 0:
                continue
            v = vals[e]
            if self.s_diag_row[e] >= 0:
                v += self.regD[self.s_diag_row[e]]
            if self.fa_entry[e]:                    # k_front_assemble forms this panel: the value goes in with the tile that holds it
                self._svals[e] = v
                continue
            self.Lval[self.s_target[e]] = v
        # storage of the panels k_front_assemble forms starts as NaN: nothing may rely on a zero-fill there
        for s_ in np.nonzero(self.front_fa)[0]:
            if self.local[s_] and self.loff[s_] >= 0:
                self.Lval[int(self.loff[s_]): int(self.loff[s_]) + pk_len(int(self.lda[s_]), int(self.ns[s_]))] = np.nan
        # upper fronts (symbolic.cpp step 13d): their zero-fill + assembly complete only at the WAIT_UPPER marker of their stream group -- until then the
        # storage holds NaN here, so that a launch that touches it too early poisons the factor
        self._upper_pending = {}
        for s_ in np.nonzero(self.front_upper)[0]:
            a, b = int(self.loff[s_]), int(self.loff[s_]) + pk_len(int(self.lda[s_]), int(self.ns[s_]))
            self._upper_pending[int(s_)] = self.Lval[a:b].copy()
            self.Lval[a:b] = np.nan
        self.U = {}
        self.fail_col = None
        self.chain_cnt[:] = 0                  # one memset per update! zeroes the tickets and completion counters
        for s_ in np.nonzero(self.single & (self.local != 0))[0]:      # k_single_factor
            d = self.Lval[self.loff[s_]]
            sj = self.sign[self.col0[s_]] if self.k2 else 1.0
            if not sj * d > 0:
                col = int(self.col0[s_])
                self.fail_col = col if self.fail_col is None else min(self.fail_col, col)
                d = 1.0
            self.Lval[self.loff[s_]] = np.sqrt(abs(d))
        self._resume = self._run(self.factor_launches, True)
        if not stop_at_marker:
            self.update_finish()

    def root_panel(self):
        """View of the root (linking) panel: what tlpk_root_panel exposes for the all-reduce."""
        s = self.root_front
        if s < 0:
            return self.Lval[:0]
        return self.panel(s).reshape(-1)          # a view of the working copy: reduced in place (the same shape on every rank)

    def update_finish(self):
        self._run(self.factor_launches, False, start=self._resume)
        assert not self._upper_pending, "upper fronts without a WAIT_UPPER marker"

    def _run(self, launches, stop_at_marker=False, start=0):
        for li in range(start, len(launches)):
            kind, first, count = (int(x) for x in launches[li])
            if kind == LK["ALLREDUCE"]:
                if stop_at_marker:
                    return li + 1
                continue
            if kind in (LK["SIDE_FORK"], LK["SIDE_JOIN"]):      # stream markers: no work
                continue
            if kind == LK["WAIT_UPPER"]:                         # `first` = stream group of the marker (-1: the fronts after the join)
                for s_ in [q for q in self._upper_pending if first < 0 or self.front_group[q] == first]:
                    assert s_ not in self._P, "an upper panel was touched before its WAIT_UPPER marker"
                    vals, a = self._upper_pending.pop(s_), int(self.loff[s_])
                    self.Lval[a: a + len(vals)] = vals
                continue
            if kind == LK["CHAIN"]:
                self._chain(self.chain_items[first: first + count])
                continue
            if kind == LK["POTRF_WIDE"]:
                kind = LK["POTRF"]
            if kind == LK["POTRF_SMALL"]:                        # count = workgroups of 4 fronts (list padded with -1)
                T = self.tasks[LK["POTRF"]][first: first + 4 * count]
                assert len(T) == 4 * count and all(int(self.ns[t[0]]) <= 16 and t[1] == 0 for t in T if t[0] >= 0)
                self._k1(T[T[:, 0] >= 0])
                continue
            if kind in (LK["FWD_SMALL"], LK["BWD_SMALL"]):      # count = workgroups of 4 fronts (list padded with -1)
                T = self.tasks[kind][first: first + 4 * count]
                assert len(T) == 4 * count
                self._small_groups(kind, T)
                continue
            if kind == LK["UPDATE_T64"]:                         # the last partial round of the preceding update launch as 64 x 64 tiles: same task array
                T = self.tasks[LK["UPDATE"]][first: first + count]
                assert (T[:, 10] == 1).all() and (T[:, 7] == 0).all()
                self._k3(T)
                continue
            if kind == LK["TRSM_THIN"]:                          # same tasks, 256 rows per task
                self._k2(self.tasks[LK["TRSM"]][first: first + count], rows_per_task=256)
                continue
            T = self.tasks[kind][first: first + count]
            getattr(self, "_k%d" % kind)(T)
        return len(launches)

    def _small_groups(self, kind, T):
        """Small fronts (<= 16 pivot columns), one wave per front: the bodies of k_fwd_small / k_bwd_small -- also the small-front items of a merged sweep launch."""
        for front, _, nb, *_r in T:
            if front < 0:
                continue
            f, ns = int(self.f[front]), int(self.ns[front])
            assert nb == ns and ns <= 16
            if kind == LK["FWD_SMALL"]:
                self._k5(np.array([[front, 0, ns, 0, 0, 0]]))
                for r0 in range(ns, f, 256):
                    self._k6(np.array([[front, 0, ns, r0, 0, 0]]))
            else:
                self._k7(np.array([[front, 0, ns, ns, f - ns, 1]]))

    def _chain(self, items):
        """k_chain: the items of one dependency-driven launch, executed in TICKET order.  Every wait must already be satisfied when its item's turn comes:
        an item may only depend on items with smaller tickets -- the property that makes the device kernel deadlock-free under any scheduling."""
        cnt = self.chain_cnt
        items = [tuple(int(v) for v in it) for it in items]

        def ready(it):
            _r, _t, _s, w0, n0, need0, w1, n1, need1, w2, need2, _sig = it
            if _r == 2:
                # early entry (TrsmTask.pad2 > 0): the strip waits INSIDE its role for the counter of its block column's diagonal block, which that role raises on
                # its way (3 x) and at its end (+ 1).  The emulator runs roles whole: the strip is runnable once the counter is final.
                pad2 = int(self.trsm_early[_t])
                if pad2 > 0 and cnt[pad2 - 1] < 4:
                    return False
            return all(cnt[w0 + q] >= need0 for q in range(n0)) and all(cnt[w1 + q] >= need1 for q in range(n1)) and (w2 < 0 or cnt[w2] >= need2)
        rng = getattr(self, "chain_rng", None)
        if rng is not None:
            # adversarial schedule: any item whose counters have arrived may run next (what the device may do with many workgroups in flight).  The
            # result must not depend on it: the waits alone order every pair of items that touch the same data.
            pending, order = list(range(len(items))), []
            while pending:
                cand = [q for q in pending[:64] if ready(items[q])]      # (a window of tickets: workgroups draw them in order)
                assert cand, "no runnable chain item"
                q = cand[int(rng.integers(len(cand)))]
                pending.remove(q); order.append(q)
                self._chain_run(items[q])
            return
        for it in items:
            assert ready(it), "chain item waits for an item with a larger ticket"
            self._chain_run(it)

    def _chain_run(self, it):
        cnt = self.chain_cnt
        for role, task, sub, w0, n0, need0, w1, n1, need1, w2, need2, sig in (it,):
            if role == 0:
                self._k3(self.tasks[LK["UPDATE"]][task: task + 1])
            elif role == 1:
                self._k1(self.tasks[LK["POTRF"]][task: task + 1])
            elif role == 2:
                self._k2(self.tasks[LK["TRSM"]][task: task + 1])
            else:
                assert role == 3 and 0 <= sub < 8
                self._k13(self.tasks[LK["UPDATE_REDUCE"]][task: task + 1], sub=sub)
            if sig >= 0:
                cnt[sig] += 4 if (role == 1 and sub == 1) else 1      # (a diagonal block with early strips: three signals on its way + the final one)

    def _k0(self, T):      # extend-add
        # group tasks by front: emulation processes whole columns ranges, children in order
        for front, j0, j1, bidx, br0, br1 in T:
            f, ns = int(self.f[front]), int(self.ns[front])
            rs = f - ns
            if front not in self.U:
                self.U[front] = np.full((rs, rs), np.nan)       # NaN = never written: catches missing tasks
            Up = self.U[front]
            P = self.panel(front)
            for c in self.kids(front):
                relc = self.relidx(c)
                rsc = len(relc)
                Uc = self.U[c]
                # the device finds the child's columns of the range in the analyse phase's lookup table
                q0, q1 = (int(v) for v in self.ea_tab[self.eatab[c] + bidx: self.eatab[c] + bidx + 2])
                assert (q0, q1) == (np.searchsorted(relc, j0), np.searchsorted(relc, j1)), "extend-add lookup table"
                assert q1 - q0 <= 16
                rlo, rhi = (int(self.ea_tab[self.eatab[c] + br0]), int(self.ea_tab[self.eatab[c] + br1])) if br1 else (0, rsc)      # row band of the task
                for q in range(q0, q1):
                    tc = relc[q]
                    lo = max(q, rlo)
                    if lo >= rhi:
                        continue
                    src = Uc[lo:rhi, q]
                    tr = relc[lo:rhi]
                    if tc < ns:
                        P[tr, tc] += src
                    else:
                        Up[tr - ns, tc - ns] += src

    def _k20(self, T):     # front assembly: tile = S entries + children (child order), written whole
        FA_CW = 4
        fa_e = np.nonzero(self.fa_entry & (self.s_target >= 0))[0]
        fa_c = self.col_of_entry[fa_e]
        for front, bc, br0, br1 in T:
            f, ns, col0 = int(self.f[front]), int(self.ns[front]), int(self.col0[front])
            npan = (ns + FA_CW - 1) // FA_CW

            def bound(k):
                return k * FA_CW if k < npan else min(f, ns + (k - npan) * FA_CW)
            j0, j1 = bc * FA_CW, min(bc * FA_CW + FA_CW, ns)
            i0, i1 = bound(br0), bound(br1)
            assert 0 < i1 - i0 <= 576 * FA_CW and j0 < ns          # FA_RB * FA_CW rows (tlpk_host.hpp)
            tile = np.zeros((i1 - i0, j1 - j0))
            loff, lda = int(self.loff[front]), int(self.lda[front])
            for q in np.nonzero((fa_c >= col0 + j0) & (fa_c < col0 + j1))[0]:      # entries of S in the tile
                e = fa_e[q]
                tc = int(fa_c[q]) - col0
                pos = int(self.s_target[e]) - loff - pk_off(lda, tc)
                if i0 <= pos < i1:
                    tile[pos - i0, tc - j0] = self._svals[e]
            for ch in self.kids(front):                                            # children, in child order
                rel = self.relidx(ch)
                tab = self.ea_tab[self.eatab[ch]:]
                q0, q1, r0, r1 = int(tab[bc]), int(tab[bc + 1]), int(tab[br0]), int(tab[br1])
                assert (q0, q1, r0, r1) == tuple(int(np.searchsorted(rel, v)) for v in (j0, j0 + FA_CW if bc + 1 < npan else ns, i0, i1)), "lookup table"
                Uc = self.U[ch]
                for q in range(q0, q1):
                    rr = np.arange(max(r0, q), r1)
                    if rr.size:
                        tile[rel[rr] - i0, rel[q] - j0] += Uc[rr, q]
            rfirst = max(i0, (j0 >> 6) << 6)
            P = self.panel(front)
            P[rfirst:i1, j0:j1] = tile[rfirst - i0:, :]

    def _k1(self, T):      # potrf of a block column's diagonal block (nb <= 256)
        for front, k0, nb, _ in T:
            P = self.panel(front)
            blk = np.tril(P[k0:k0 + nb, k0:k0 + nb])
            sg = self.sign[self.col0[front] + k0: self.col0[front] + k0 + nb] if self.k2 else np.ones(nb)
            for j in range(nb):
                d = blk[j, j]
                if not sg[j] * d > 0:
                    col = int(self.col0[front] + k0 + j)
                    self.fail_col = col if self.fail_col is None else min(self.fail_col, col)
                    d = sg[j]
                blk[j + 1:, j + 1:] -= np.tril(np.outer(blk[j + 1:, j], blk[j + 1:, j]) / d)
                blk[j, j] = np.sqrt(abs(d))
                blk[j + 1:, j] /= sg[j] * blk[j, j]                    # K = L S L': L_ij = a_ij / (s_j L_jj)
            P[k0:k0 + nb, k0:k0 + nb] = blk

    def _k2(self, T, rows_per_task=64):      # trsm: rows below the diagonal block of a block column, whole width
        import scipy.linalg as sla
        for front, k0, nb, row0, _kprev, rowlim in T:
            P = self.panel(front)
            f = int(self.f[front])
            r1 = int(rowlim)                                      # rows [row0, rowlim) belong to the task
            assert row0 >= k0 + nb and row0 < r1 <= f and r1 - row0 <= rows_per_task
            L11 = np.tril(P[k0:k0 + nb, k0:k0 + nb])
            X = sla.solve_triangular(L11, P[row0:r1, k0:k0 + nb].T, lower=True).T
            if self.k2:
                X = X * self.sign[self.col0[front] + k0: self.col0[front] + k0 + nb][None, :]     # L21 = A21 L11^-T S11
            P[row0:r1, k0:k0 + nb] = X

    def _k3(self, T):      # update
        for front, k0, kw, i0, j0, jlim, beta0, slot1, seg, nsl, t64 in T:
            TILE = {0: 128, 1: 64, 2: 32}[int(t64)]      # UpdateTask.pad2: 1 = a 64 x 64 tile, 2 = a 32 x 32 tile of a diagonal block's short update
            assert not (t64 and slot1), "the small tiles are never split-K parts"
            P = self.panel(front)
            f, ns = int(self.f[front]), int(self.ns[front])
            i1, j1 = min(i0 + TILE, f), min(j0 + TILE, jlim)
            if i1 <= i0 or j1 <= j0:
                continue
            if seg:
                # skip list: only the listed K slabs (+ the partial last slab) are multiplied, exactly as k_update does
                nseg = int(self.upd_seg[seg - 1])
                kcols = [np.arange(self.upd_seg[seg + 2 * q], self.upd_seg[seg + 2 * q] + 16 * self.upd_seg[seg + 2 * q + 1]) for q in range(nseg)]
                kcols.append(np.arange(k0 + (kw // 16) * 16, k0 + kw))
                kcols = np.concatenate(kcols)
                assert kcols.size == 16 * nsl + kw % 16 and (nsl >= 2 or t64) and kcols.min() >= k0 and kcols.max() < k0 + kw and np.all(np.diff(kcols) > 0)
            else:
                kcols = np.arange(k0, k0 + kw)
            Pj = P[j0:j1][:, kcols]
            if self.k2:
                Pj = Pj * self.sign[self.col0[front] + kcols][None, :]      # X S X'
            if getattr(self, "exact_k_order", False):
                # one multiply-add per K column and entry, in K order: the sum of an entry does not depend on the SHAPE of the tile that holds it (a BLAS
                # product rounds differently for a 64-row and a 128-row operand) -- for tests that compare schedules with different tile shapes bit for bit
                Pi = P[i0:i1][:, kcols]
                G = np.zeros((i1 - i0, j1 - j0))
                for q in range(len(kcols)):
                    G += Pi[:, q:q + 1] * Pj[:, q][None, :]
            else:
                G = P[i0:i1][:, kcols] @ Pj.T
            if slot1:            # split-K part: raw tile to its scratch slot (each slot written exactly once)
                assert int(slot1) - 1 not in self.spart
                self.spart[int(slot1) - 1] = G
                continue
            rr = np.arange(i0, i1)[:, None]; cc = np.arange(j0, j1)[None, :]
            mask = rr >= cc
            for c in range(j0, j1):
                rsel = np.arange(max(i0, c), i1)
                if rsel.size == 0:
                    continue
                if c < ns:
                    P[rsel, c] -= G[rsel - i0, c - j0]
                else:
                    if front not in self.U:
                        self.U[front] = np.full((f - ns, f - ns), np.nan)
                    if beta0:
                        self.U[front][rsel - ns, c - ns] = -G[rsel - i0, c - j0]
                    else:
                        self.U[front][rsel - ns, c - ns] -= G[rsel - i0, c - j0]
            del mask

    def _k13(self, T, sub=None):     # split-K reduce: parts of a tile summed in slot order, then applied like _k3
        TILE = 128                   # sub = one eighth of the tile's columns (a chain item); None = the whole tile (a launch: eight workgroups)
        for front, slot0, parts, i0, j0, jlim, beta0, _ in T:
            f, ns = int(self.f[front]), int(self.ns[front])
            i1, j1 = min(i0 + TILE, f), min(j0 + TILE, jlim)
            take = (lambda q: self.spart.pop(q)) if sub is None else (lambda q: self.spart[q])     # pop: a slot may be reused by a later launch of the stream (never inside a chain launch)
            G = take(int(slot0))
            for sp in range(1, int(parts)):
                G = G + take(int(slot0) + sp)
            P = self.panel(front)
            for c in (range(j0, j1) if sub is None else range(j0 + 16 * sub, min(j0 + 16 * sub + 16, j1))):
                rsel = np.arange(max(i0, c), i1)
                if rsel.size == 0:
                    continue
                if c < ns:
                    P[rsel, c] -= G[rsel - i0, c - j0]
                else:
                    if front not in self.U:
                        self.U[front] = np.full((f - ns, f - ns), np.nan)
                    if beta0:
                        self.U[front][rsel - ns, c - ns] = -G[rsel - i0, c - j0]
                    else:
                        self.U[front][rsel - ns, c - ns] -= G[rsel - i0, c - j0]

    # ---- solve! ----
    def solve_local(self, xi_p, xi_d, A, rhs_rank=None):
        rank = self.rank if rhs_rank is None else rhs_rank                  # refinement: every rank adds its partial residual on linking rows
        if self.k2:                                                     # k_k2_rhs: [xi_d ; xi_p] permuted; sharded: own nodes, root nodes on rank 0
            nl = self.row_local[self.perm]
            self.xw = np.where((nl == 0) | ((nl == 2) & (self.rank != 0)), 0.0, np.concatenate([xi_d, xi_p])[self.perm])
        else:
            # k_rhs: a rank sums only its own columns; only rank 0 adds xi_p on linking rows
            Aloc = A @ __import__("scipy.sparse").sparse.diags(self.col_local.astype(float))
            xi = Aloc @ (self.D * xi_d)
            rl = self.row_local
            xi = np.where(rl == 0, 0.0, xi + np.where((rl == 2) & (rank != 0), 0.0, xi_p))
            self.xw = xi[self.perm].copy()
        for s_ in np.nonzero(self.single & (self.local != 0))[0]:          # k_single_solve
            l = self.Lval[self.loff[s_]]
            self.xw[self.col0[s_]] = self.xw[self.col0[s_]] / l / l
        self.ucflat = np.full(max(int((self.ucoff + self.f - self.ns).max()), 1), np.nan)
        self._resume_fwd = self._run(self.fwd_launches, True)

    def root_rhs(self):
        s = self.root_front
        if s < 0:
            return self.xw[:0]
        return self.xw[self.col0[s]: self.col0[s] + self.ns[s]]

    def solve_finish(self, xi_d, A):
        self._run(self.fwd_launches, False, start=self._resume_fwd)
        self._bwd_seen = {}
        if self.k2:
            self.xw *= self.sign                                        # k_apply_signs: L S L' x = b
        self._run(self.bwd_launches)
        if self.k2:                                                     # k_k2_out
            sol = np.zeros(self.m + self.n)
            sol[self.perm] = np.where(self.row_local[self.perm] != 0, self.xw, 0.0)
            return sol[: self.n], sol[self.n:]
        dy = np.zeros(self.m)
        dy[self.perm] = self.xw
        dy[self.row_local == 0] = 0.0
        dx = np.where(self.col_local != 0, self.D * (A.T @ dy - xi_d), 0.0)
        return dx, dy

    # one iterative-refinement step in two halves (tlpk_refine_local / tlpk_refine_finish): residuals of the rows / columns this rank
    # owns, partial sums on the linking rows (dx is zero outside the rank's columns; rank 0 adds xi_p - Rd dy there)
    def refine_local(self, dx, dy, xi_p, xi_d, A, theta, regP, regD):
        assert not self.k2
        rl = self.row_local
        base = np.where((rl == 2) & (self.rank != 0), 0.0, xi_p - regD * dy)
        r1 = base - A @ dx
        self._r2 = (xi_d + (theta + regP) * dx) - A.T @ dy
        self.solve_local(r1, self._r2, A, rhs_rank=0)

    def refine_finish(self, dx, dy, A):
        cx, cy = self.solve_finish(self._r2, A)
        return dx + cx, dy + cy

    def solve(self, xi_p, xi_d, A):
        self.solve_local(xi_p, xi_d, A)
        return self.solve_finish(xi_d, A)

    def _k4(self, T):      # fwd gather: one row chunk of a front, sources from the gather lists
        for front, _, nrows, row0, *_ in T:
            f, ns = int(self.f[front]), int(self.ns[front])
            c0 = int(self.col0[front]); ro = int(self.rowoff[front]); uo = int(self.ucoff[front])
            for r in range(row0, min(row0 + int(nrows), f)):
                q0, q1 = self.gth_ptr[ro + r], self.gth_ptr[ro + r + 1]
                v = self.xw[c0 + r] if r < ns else 0.0
                for q in range(q0, q1):
                    assert not np.isnan(self.ucflat[self.gth_src[q]])
                    v += self.ucflat[self.gth_src[q]]
                if r < ns:
                    self.xw[c0 + r] = v
                else:
                    self.ucflat[uo + r - ns] = v

    def _k5(self, T):      # fwd diag
        import scipy.linalg as sla
        for front, k0, nb, *_ in T:
            P = self.panel(front); c0 = int(self.col0[front])
            L11 = np.tril(P[k0:k0 + nb, k0:k0 + nb])
            self.xw[c0 + k0: c0 + k0 + nb] = sla.solve_triangular(L11, self.xw[c0 + k0: c0 + k0 + nb], lower=True)

    def _k6(self, T):      # fwd update
        for front, k0, nb, row0, *_ in T:
            P = self.panel(front); c0 = int(self.col0[front])
            f, ns = int(self.f[front]), int(self.ns[front])
            r1 = min(row0 + 256, f)
            acc = P[row0:r1, k0:k0 + nb] @ self.xw[c0 + k0: c0 + k0 + nb]
            uo = int(self.ucoff[front])
            for t, r in enumerate(range(row0, r1)):
                if r < ns:
                    self.xw[c0 + r] -= acc[t]
                else:
                    self.ucflat[uo + r - ns] -= acc[t]
            if _[1] > 0:                      # fused look-ahead: solve the next diagonal block
                assert row0 == k0 + nb and r1 >= min(k0 + nb + _[1], ns)
                self._k5(np.array([[front, k0 + nb, _[1], 0, 0, 0]]))

    def _k7(self, T):      # bwd step: remove solved source rows from one column block (+ solve it)
        import scipy.linalg as sla
        for front, k0, nb, row0, nrows, diag in T:
            P = self.panel(front); c0 = int(self.col0[front])
            f, ns = int(self.f[front]), int(self.ns[front])
            rows = self.rows(front)
            assert row0 >= k0 + nb and row0 + nrows <= f
            if nrows > 0:
                xf = np.concatenate([self.xw[c0: c0 + ns], self.xw[rows[ns:]]])
                self.xw[c0 + k0: c0 + k0 + nb] -= P[row0:row0 + nrows, k0:k0 + nb].T @ xf[row0:row0 + nrows]
            self._bwd_seen.setdefault((int(front), int(k0)), []).append((int(row0), int(nrows)))
            if diag:
                # every row below the block must have been applied exactly once before it is solved
                seen = sorted(self._bwd_seen[(int(front), int(k0))])
                pos = k0 + nb
                for r0, nr in seen:
                    assert r0 == pos or nr == 0, (front, k0, seen)
                    pos = max(pos, r0 + nr)
                assert pos == f, (front, k0, seen)
                L11 = np.tril(P[k0:k0 + nb, k0:k0 + nb])
                self.xw[c0 + k0: c0 + k0 + nb] = sla.solve_triangular(L11.T, self.xw[c0 + k0: c0 + k0 + nb], lower=False)

    # Persistent sweeps: the items of a launch are executed in LIST order = ticket order.  The device hands
    # them to workgroups in that order, and an item may only wait for items with a smaller ticket (otherwise
    # the sweep could deadlock): every block an item consumes must already be published when its turn comes.
    def _k18(self, T, SW=64):     # forward sweep: SWEEP_NB-wide pivot blocks
        import scipy.linalg as sla
        pub = set()
        for front, k0, nb, _r0, pivot, nin in T:
            if pivot == 2:                                        # a group of four small fronts riding in the sweep's launch (round 6)
                self._small_groups(LK["FWD_SMALL"], self.tasks[LK["FWD_SMALL"]][4 * k0: 4 * k0 + 4])
                continue
            P = self.panel(front); c0 = int(self.col0[front])
            f, ns = int(self.f[front]), int(self.ns[front])
            assert self.flagoff[front] >= 0
            acc = np.zeros(nb)
            for j in range(nin):                                  # consumed blocks, ascending
                assert (int(front), j) in pub, ("forward sweep item waits for a later ticket", front, k0, j)
                w = min(SW, ns - SW * j)
                acc += P[k0:k0 + nb, SW * j:SW * j + w] @ self.xw[c0 + SW * j: c0 + SW * j + w]
            if pivot:
                assert k0 % SW == 0 and nin == k0 // SW and k0 + nb <= ns and nb <= SW
                rhs = self.xw[c0 + k0: c0 + k0 + nb] - acc
                L11 = np.tril(P[k0:k0 + nb, k0:k0 + nb])
                self.xw[c0 + k0: c0 + k0 + nb] = sla.solve_triangular(L11, rhs, lower=True)
                pub.add((int(front), k0 // SW))
            else:
                assert k0 >= ns and nin == (ns + SW - 1) // SW and nb <= 128
                uo = int(self.ucoff[front])
                self.ucflat[uo + k0 - ns: uo + k0 - ns + nb] -= acc

    def _k19(self, T, SW=64):     # backward sweep
        import scipy.linalg as sla
        pub = set()
        for front, k0, nb, row0, nrows, nlater in T:
            if nlater == -2:                                      # a group of four small fronts riding in the sweep's launch (round 6)
                self._small_groups(LK["BWD_SMALL"], self.tasks[LK["BWD_SMALL"]][4 * k0: 4 * k0 + 4])
                continue
            P = self.panel(front); c0 = int(self.col0[front])
            f, ns = int(self.f[front]), int(self.ns[front])
            rows = self.rows(front)
            nblk = (ns + SW - 1) // SW
            assert row0 == ns and nrows == f - ns and k0 % SW == 0 and nlater == nblk - 1 - k0 // SW and nb <= SW
            acc = np.zeros(nb)
            if nrows > 0:                                         # rows below the pivot block: ancestors' values
                acc += P[ns:f, k0:k0 + nb].T @ self.xw[rows[ns:]]
            for q in range(nlater):                               # later blocks, last first
                j = nblk - 1 - q
                assert (int(front), j) in pub, ("backward sweep item waits for a later ticket", front, k0, j)
                w = min(SW, ns - SW * j)
                acc += P[SW * j:SW * j + w, k0:k0 + nb].T @ self.xw[c0 + SW * j: c0 + SW * j + w]
            L11 = np.tril(P[k0:k0 + nb, k0:k0 + nb])
            self.xw[c0 + k0: c0 + k0 + nb] = sla.solve_triangular(L11.T, self.xw[c0 + k0: c0 + k0 + nb] - acc, lower=False)
            pub.add((int(front), k0 // SW))

    # dense L in permuted numbering, from the panels
    def dense_L(self):
        L = np.zeros((self.m, self.m))
        for s in range(len(self.f)):
            if not self.local[s]:
                continue
            P = self.panel(s); rows = self.rows(s); ns = int(self.ns[s]); c0 = int(self.col0[s])
            for c in range(ns):
                L[rows[c:], c0 + c] = P[c:, c]
        return L


def pk_off(lda, col):
    """Offset of the virtual row 0 of panel column `col` (tlpk_host.hpp)."""
    b = col >> 6
    return col * lda - 64 * b * (col - 32 * b - 31)


def pk_len(lda, ns):
    return pk_off(lda, ns) + 64 * (ns >> 6)


def unpack_panel(lval, loff, f, ns, lda):
    """f x ns array of a panel stored by 64-column slices; the never-stored blocks above the diagonal blocks read as 0."""
    P = np.zeros((f, ns))
    for b in range((ns + 63) // 64):
        w = min(64, ns - 64 * b); ld = lda - 64 * b
        start = loff + pk_off(lda, 64 * b) + 64 * b
        blk = lval[start: start + w * ld].reshape((ld, w), order="F")
        P[64 * b:, 64 * b: 64 * b + w] = blk[: f - 64 * b]
    return P


def panels_to_dense_L(kkt, lval):
    """Dense L (permuted numbering) from panel storage copied off the device."""
    em = Emulator.__new__(Emulator)
    g = kkt.symbolic
    em.m = kkt.m
    em.f = g("front_f"); em.ns = g("front_ns"); em.col0 = g("front_col0"); em.loff = g("front_loff"); em.lda = g("front_lda")
    em.rowoff = g("front_rowoff"); em.rowidx = g("rowidx"); em.local = g("front_local")
    em.Lval = lval
    return em.dense_L()
