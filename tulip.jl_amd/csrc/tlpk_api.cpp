// tlpk_api.cpp -- the C ABI of libtlpk.so (include/tlpk.h): handle life cycle, uploads, and
// the replay of the static launch schedules on the handle's HIP stream.
//
// Mirrors the solver object of the reference backend
// (/root/reference/src/KKT/Cholmod/cholmod.jl:46-60: m, n, A, theta, regP, regD, K, F, xi) with
// device-resident state: A (CSC+CSR), stored copies of theta/regP/regD, the supernodal factor.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <condition_variable>
#include <functional>
#include <mutex>

#include "../../include/tlpk.h"
#include "tlpk_device.hpp"
#include "tlpk_handle.hpp"
#include "hostcopy.hpp"

using namespace tlpk;

namespace {

// GPU_MAX_HW_QUEUES (more hardware queues than the runtime's default 4) is a tuning knob of the HOST
// process: the Python and Julia glue set it before the HIP runtime initialises (tulip.jl_amd/__init__.py,
// julia/libtlpk.jl).  The library itself never touches the environment.

// One persistent host thread per shard of a multi-device handle (round 4).  A Newton step of one shard is ~220 + 4 x 40 launches; enqueued by ONE
// thread, shard after shard, eight shards cost the host ~8 x 1.5 ms against a device budget of ~14 ms per step (round-3 review).  The calls below
// hand the per-shard part of a phase to the pool: shard 0 runs on the calling thread, shard r on worker r - 1; a phase ends when all have returned
// (the reductions between the phases are enqueued by the caller).  Every worker sets its shard's device itself (hipSetDevice is per thread).
// TLPK_SHARD_THREADS=0: the caller runs all shards in turn (the round-3 behaviour; for the A/B in profiles/).
struct ShardPool {
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv, cv_done;
    const std::function<int(int)> *job = nullptr;
    unsigned long long gen = 0; int pending = 0, n = 0; bool stop = false; std::vector<int> rcs;
    void worker(int r) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<int(int)> *f;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || gen != seen; }); if (stop) return; seen = gen; f = job; }
            int rc;
            try { rc = (*f)(r); } catch (...) { rc = TLPK_INTERNAL; }
            { std::lock_guard<std::mutex> lk(mu); rcs[(size_t)r] = rc; if (--pending == 0) cv_done.notify_one(); }
        }
    }
    explicit ShardPool(int nshards) : n(nshards), rcs((size_t)nshards, TLPK_OK) {
        try { for (int r = 1; r < nshards; ++r) th.emplace_back(&ShardPool::worker, this, r); } catch (...) { /* fewer workers: run() covers the rest itself */ }
    }
    ~ShardPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto &t : th) t.join(); }
    // fn(r) for every shard; returns the result of the lowest-numbered shard that failed (and its index in *who)
    int run(const std::function<int(int)> &fn, int *who) {
        const int nw = (int)th.size();
        { std::lock_guard<std::mutex> lk(mu); job = &fn; pending = nw; ++gen; }
        if (nw) cv.notify_all();
        int rc0;
        try { rc0 = fn(0); } catch (...) { rc0 = TLPK_INTERNAL; }
        rcs[0] = rc0;
        for (int r = nw + 1; r < n; ++r) { try { rcs[(size_t)r] = fn(r); } catch (...) { rcs[(size_t)r] = TLPK_INTERNAL; } }      // shards without a worker
        if (nw) { std::unique_lock<std::mutex> lk(mu); cv_done.wait(lk, [&] { return pending == 0; }); }
        for (int r = 0; r < n; ++r) if (rcs[(size_t)r] != TLPK_OK) { if (who) *who = r; return rcs[(size_t)r]; }
        return TLPK_OK;
    }
};
// fn(shard handle, index) for every shard of a multi-device handle, concurrently; on failure the shard's message becomes the parent's
int for_shards(tlpk_handle *h, const std::function<int(tlpk_handle *, int)> &fn) {
    const char *te = std::getenv("TLPK_SHARD_THREADS");      // (read per call: tests and A/B runs switch it inside one process)
    const bool threads = !te || std::atoi(te) != 0;
    const int N = (int)h->sub.size();
    auto one = [&](int r) -> int {
        tlpk_handle *c = h->sub[(size_t)r];
        if (hipSetDevice(c->device) != hipSuccess) { c->last_error = "hipSetDevice failed"; return TLPK_HIPERR; }
        return fn(c, r);
    };
    int who = -1, rc = TLPK_OK;
    if (threads && N > 1) {
        if (!h->shard_pool) h->shard_pool = new ShardPool(N);
        const std::function<int(int)> f = one;
        rc = static_cast<ShardPool *>(h->shard_pool)->run(f, &who);
    } else {
        for (int r = 0; r < N && rc == TLPK_OK; ++r) { rc = one(r); if (rc != TLPK_OK) who = r; }
    }
    if (rc != TLPK_OK && who >= 0) h->last_error = h->sub[(size_t)who]->last_error;
    (void)hipSetDevice(h->sub[0]->device);
    return rc;
}

void shard_pool_delete(void *p) { delete static_cast<ShardPool *>(p); }

int kind_class(i32 kind) {
    switch (kind) {
    case LK_EXTEND_ADD: case LK_FRONT_ASSEMBLE: return TLPK_KC_EXTEND_ADD;
    case LK_POTRF: case LK_POTRF_WIDE: case LK_POTRF_SMALL: return TLPK_KC_POTRF;
    case LK_TRSM: case LK_TRSM_THIN: return TLPK_KC_TRSM;
    case LK_UPDATE: case LK_UPDATE_T64: return TLPK_KC_UPDATE;
    case LK_UPDATE_REDUCE: return TLPK_KC_UPDATE_REDUCE;
    case LK_CHAIN: return TLPK_KC_CHAIN;
    case LK_FWD_GATHER: case LK_FWD_DIAG: case LK_FWD_UPDATE: case LK_FWD_SMALL: case LK_FWD_SWEEP: return TLPK_KC_SOLVE_FWD;
    default: return TLPK_KC_SOLVE_BWD;
    }
}

// profile helpers: record an event pair around one launch
struct ProfScope {
    tlpk_handle *h; bool on; size_t idx; hipStream_t st;
    ProfScope(tlpk_handle *h_, int cls, hipStream_t st_ = nullptr, const Launch *L = nullptr) : h(h_), on(h_->profile), idx(0), st(st_ ? st_ : h_->stream) {
        if (!on) return;
        if (h->ev_used + 2 > h->ev_pool.size()) {
            const size_t old = h->ev_pool.size();
            h->ev_pool.resize(old + 512);
            for (size_t i = old; i < h->ev_pool.size(); ++i) hipEventCreate(&h->ev_pool[i]);
        }
        idx = h->ev_used; h->ev_used += 2;
        h->ev_class.push_back(cls);
        h->ev_launch.push_back(L ? std::array<long long, 3>{L->kind, L->first, L->count} : std::array<long long, 3>{-1, 0, 0});
        hipEventRecord(h->ev_pool[idx], st);
    }
    ~ProfScope() { if (on) hipEventRecord(h->ev_pool[idx + 1], st); }
};
void prof_begin(tlpk_handle *h, bool reset) {
    if (!h->profile) return;
    // pending (not yet collected) event pairs of earlier async solves stay queued: the pool grows
    // until the next prof_collect, which runs after a stream synchronisation
    if (reset) { std::memset(&h->kt, 0, sizeof(h->kt)); h->ev_used = 0; h->ev_class.clear(); h->ev_launch.clear(); }
}
void prof_collect(tlpk_handle *h) {          // stream must be synchronised
    if (!h->profile) return;
    // TLPK_PROF_DUMP=<file>: one line per timed launch (class, kind, first task, task count, ms) -- tools/update_launch_eff.py
    const char *dump = std::getenv("TLPK_PROF_DUMP");
    FILE *df = (dump && *dump) ? std::fopen(dump, "a") : nullptr;
    for (size_t i = 0; i < h->ev_class.size(); ++i) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]);
        h->kt.ms[h->ev_class[i]] += ms;
        if (h->ev_launch[i][0] != LK_UPDATE_T64) h->kt.launches[h->ev_class[i]] += 1;      // (the 64 x 64 tail of an update launch counts as part of that launch)
        if (df) std::fprintf(df, "%d %lld %lld %lld %.6f\n", h->ev_class[i], h->ev_launch[i][0], h->ev_launch[i][1], h->ev_launch[i][2], (double)ms);
    }
    if (df) { std::fprintf(df, "#\n"); std::fclose(df); }
    h->ev_used = 0; h->ev_class.clear(); h->ev_launch.clear();
}

// Stream groups: launches tagged with group g >= 1 go to their own stream.  fork = the group
// streams wait for everything enqueued on the main stream so far; join = the main stream waits for
// the group streams.  A launch tagged -1 (depth-0 fronts) and the end of a range force a join.
void fork_groups(tlpk_handle *h) {
    if (h->forked || h->S.ngroups <= 1) return;
    hipEventRecord(h->ev_fork, h->stream);
    for (int g = 1; g < h->S.ngroups; ++g) hipStreamWaitEvent(h->gstream[g], h->ev_fork, 0);
    h->forked = true;
}
void join_groups(tlpk_handle *h) {
    if (!h->forked) return;
    for (int g = 1; g < h->S.ngroups; ++g) {
        hipEventRecord(h->ev_join[g], h->gstream[g]);
        hipStreamWaitEvent(h->stream, h->ev_join[g], 0);
    }
    h->forked = false;
}

// dir: 0 = forward-solve schedule, 1 = backward-solve schedule, -1 = factorisation (no sweeps)
// base != nullptr: the launches of group -1 (the root front) go to that stream instead of the handle's main stream
void run_launches(tlpk_handle *h, const std::vector<Launch> &L, size_t from, size_t to, int dir = -1, int nrhs = 1, hipStream_t base = nullptr) {
    size_t skip_update = (size_t)-1;
    for (size_t i = from; i < to; ++i) {
        if (L[i].group < 0) join_groups(h);
        if (L[i].kind == LK_ALLREDUCE_ROOT) continue;
        hipStream_t st = (base && L[i].group < 0) ? base : h->stream;
        // profiling serialises everything on the main stream: per-launch HIP-event durations are
        // then the kernels' own durations, not time shared with other groups' kernels
        const bool marker = (L[i].kind == LK_SIDE_FORK || L[i].kind == LK_SIDE_JOIN || L[i].kind == LK_WAIT_UPPER);
        if (h->profile || h->serial) { if (marker) continue; }
        else {
            if (L[i].group >= 1) { fork_groups(h); st = h->gstream[L[i].group]; }
            else if (L[i].group == 0) fork_groups(h);      // group 0 runs on the main stream itself
            if (L[i].kind == LK_WAIT_UPPER) {              // step 13d: the upper panels are zero-filled and assembled from here on
                if (h->upper_split) hipStreamWaitEvent(st, h->ev_upper, 0);
                continue;
            }
            // side stream of the group (slot 0 also serves the depth-0 fronts, group -1)
            const int sg = std::max(L[i].group, 0);
            if (L[i].kind == LK_SIDE_FORK) {
                hipEventRecord(h->ev_side[sg], st);
                hipStreamWaitEvent(h->sstream[sg], h->ev_side[sg], 0);
                continue;
            }
            if (L[i].kind == LK_SIDE_JOIN) {
                hipEventRecord(h->ev_side[sg], h->sstream[sg]);
                hipStreamWaitEvent(st, h->ev_side[sg], 0);
                continue;
            }
            if (L[i].side) st = h->sstream[sg];
        }
        Launch cur = L[i];
        if ((h->profile || h->serial) && cur.kind == LK_UPDATE) {
            // single-stream modes: the side-stream update of a block column's diagonal tiles and the
            // main-stream update of the rows below are adjacent task ranges; run them as ONE launch
            // (before the potrf), so that a serialised launch still has the whole device to fill
            if (skip_update == i) continue;
            if (cur.side) {
                for (size_t j = i + 1; j < to && L[j].kind != LK_SIDE_JOIN; ++j)
                    if (L[j].kind == LK_UPDATE && !L[j].side && L[j].first == cur.first + cur.count) {
                        cur.count += L[j].count; skip_update = j; break;
                    }
            }
        }
        // TLPK_STAGGER=1 (experiment): group 1 starts an extend-add launch only after group 0's previous one has finished, so that the
        // HBM-bound extend-add of one group runs beside the matrix-core-bound updates of the other instead of beside ITS extend-add
        const bool stag = h->stagger && dir < 0 && !h->profile && !h->serial && cur.kind == LK_EXTEND_ADD && h->S.ngroups >= 2 && h->ev_stagger;
        if (stag && L[i].group == 1 && h->stagger_armed) { hipStreamWaitEvent(st, h->ev_stagger, 0); h->stagger_armed = false; }
        ProfScope ps(h, kind_class(cur.kind), st, &cur);
        if (stag && L[i].group == 0 && cur.count >= h->stagger_min) {
            launch_tasks(st, h->d, cur, nullptr, nrhs);
            hipEventRecord(h->ev_stagger, st); h->stagger_armed = true;
            continue;
        }
        if (cur.kind == LK_FWD_SWEEP || cur.kind == LK_BWD_SWEEP) {
            const i32 slot = (dir == 0 ? h->sweep_slot_fwd : h->sweep_slot_bwd)[i];
            SweepArgs sw{h->d.sweep_tickets + slot, h->d.sweep_xh + (dir == 0 ? 0 : h->S.m), 2 * h->S.m, h->poll[0], h->poll[1], h->poll[2], dir == 0 ? h->d.fwd_small_tasks : h->d.bwd_small_tasks};
            launch_tasks(st, h->d, cur, &sw, nrhs);
        } else
            launch_tasks(st, h->d, cur, nullptr, nrhs);
    }
    join_groups(h);
}

int upload_all(tlpk_handle *h) {
    Symbolic &S = h->S;
    DevArrays &d = h->d;
    d.m = S.m; d.n = S.n;
    int rc;
#define UP(dst, vec) if ((rc = dev_upload(h, &(dst), (vec))) != TLPK_OK) return rc
    UP(d.Ap, S.Ap); UP(d.Ai, S.Ai); UP(d.Ax, S.Ax);
    UP(d.Tp, S.Tp); UP(d.Tj, S.Tj);
    {
        std::vector<double> Tx(S.Tpos.size());
        for (size_t q = 0; q < Tx.size(); ++q) Tx[q] = S.Ax[S.Tpos[q]];
        UP(d.Tx, Tx);
    }
    UP(d.perm, S.perm);
    if (S.system == 0) {
        // a second CSR copy with the rows in PERMUTED order, for the right-hand-side kernel of every solve (k_rhs walks the permuted rows:
        // with the original row order each 8-lane group started with perm[ii] -> Tp[i] -> Tj[q] -> w[j], four dependent loads at scattered
        // addresses, 65 us = 0.6 TB/s on config C4; now Tp / Tj / Tx are streamed).  Entries keep their order inside a row: same sums.
        std::vector<i64> Pp((size_t)S.m + 1, 0);
        for (i64 ii = 0; ii < S.m; ++ii) { const i32 i = S.perm[(size_t)ii]; Pp[(size_t)ii + 1] = Pp[(size_t)ii] + (S.Tp[(size_t)i + 1] - S.Tp[(size_t)i]); }
        std::vector<i32> Pj((size_t)Pp[(size_t)S.m]); std::vector<double> Px((size_t)Pp[(size_t)S.m]);
        for (i64 ii = 0; ii < S.m; ++ii) {
            const i32 i = S.perm[(size_t)ii];
            i64 o = Pp[(size_t)ii];
            for (i64 q = S.Tp[(size_t)i]; q < S.Tp[(size_t)i + 1]; ++q, ++o) { Pj[(size_t)o] = S.Tj[(size_t)q]; Px[(size_t)o] = S.Ax[(size_t)S.Tpos[(size_t)q]]; }
        }
        UP(d.Pp, Pp); UP(d.Pj, Pj); UP(d.Px, Px);
    }
    UP(d.row_local, S.row_local); UP(d.col_local, S.col_local);
    {
        // compact the assembly lists to the entries this rank owns.  One rank (every entry local): the lists of the analyse phase ARE the compact ones -- no copies
        // (round 5: the element-by-element compaction of 5 - 30 million entries and an unconditional second copy of the target list were ~70 ms of KKT.setup on C4)
        bool every = true;
        for (i64 e = 0; e < S.nnzS && every; ++e) every = S.s_local[(size_t)e] != 0;
        std::vector<i64> tgt, ptr; std::vector<i32> diag;
        i64 np = 0;
        if (every) { d.n_asm = S.nnzS; np = S.pair_ptr[(size_t)S.nnzS]; }
        else {
            tgt.reserve((size_t)S.nnzS); diag.reserve((size_t)S.nnzS); ptr.reserve((size_t)S.nnzS + 1);
            ptr.push_back(0);
            for (i64 e = 0; e < S.nnzS; ++e) {
                if (!S.s_local[e]) continue;
                tgt.push_back(S.s_target[e]); diag.push_back(S.s_diag_row[e]);
                np += S.pair_ptr[e + 1] - S.pair_ptr[e];
                ptr.push_back(np);
            }
            d.n_asm = (i64)tgt.size();
        }
        {
            // per permuted column: its first entry in the compacted list (entries are in column order); and the target list of
            // k_assemble: -1 for the entries of the fronts whose panels k_front_assemble forms
            bool any_fa = false;
            for (char f : S.front_fa) any_fa |= (f != 0);
            std::vector<i64> colptr_c;
            if (!every) {
                colptr_c.assign((size_t)S.m + 1, 0);
                i64 cnt = 0;
                for (i64 kk = 0; kk < S.m; ++kk) {
                    colptr_c[(size_t)kk] = cnt;
                    for (i64 e = S.Sp[(size_t)kk]; e < S.Sp[(size_t)kk + 1]; ++e) cnt += (S.s_local[(size_t)e] != 0);
                }
                colptr_c[(size_t)S.m] = cnt;
                if (cnt != d.n_asm) { h->last_error = "assembly list: column pointers do not match the compacted entries"; return TLPK_INTERNAL; }
            }
            const std::vector<i64> &colptr = every ? S.Sp : colptr_c;
            if (any_fa) {      // k_front_assemble is off by default: no second copy of the target list then
                std::vector<i64> tsmall;
                if (every) tsmall.assign(S.s_target.begin(), S.s_target.end()); else tsmall = tgt;
                for (i64 kk = 0; kk < S.m; ++kk)
                    if (S.front_fa[(size_t)S.sn_of_col[(size_t)kk]])
                        for (i64 q = colptr[(size_t)kk]; q < colptr[(size_t)kk + 1]; ++q) tsmall[(size_t)q] = -1;
                UP(d.asm_colptr, colptr); UP(d.asm_target_small, tsmall);
            }
            // step 13d: which entries belong to an upper front (assembled on a stream of their own)
            bool any = false;
            for (char f : S.front_upper) any |= (f != 0);
            d.has_upper = any;
            if (any) {
                std::vector<unsigned char> up((size_t)d.n_asm, 0);
                for (i64 kk = 0; kk < S.m; ++kk)
                    if (S.front_upper[(size_t)S.sn_of_col[(size_t)kk]]) std::fill(up.begin() + colptr[(size_t)kk], up.begin() + colptr[(size_t)kk + 1], (unsigned char)1);
                UP(d.asm_upper, up);
            }
        }
        // pairs of local entries are contiguous per entry; entries of non-local fronts have none,
        // so the pair arrays are already compact and in the same order.
        if (every) { UP(d.asm_target, S.s_target); UP(d.asm_diag, S.s_diag_row); UP(d.asm_ptr, S.pair_ptr); }
        else { UP(d.asm_target, tgt); UP(d.asm_diag, diag); UP(d.asm_ptr, ptr); }
        if (!d.asm_target_small) d.asm_target_small = d.asm_target;
        UP(d.pair_w, S.pair_w); UP(d.pair_j, S.pair_j);
        if (np != (i64)S.pair_w.size()) { h->last_error = "assembly list compaction mismatch"; return TLPK_INTERNAL; }
    }
    const FrontDesc *fr = nullptr; const i32 *ri = nullptr, *re = nullptr, *ch = nullptr;
    { FrontDesc *p; UP(p, S.fronts); fr = p; }
    { i32 *p; UP(p, S.rowidx); ri = p; }
    { i32 *p; UP(p, S.rel); re = p; }
    { i32 *p; UP(p, S.ea_tab); d.ctx.ea_tab = p; }
    { i32 *p; UP(p, S.children); ch = p; }
    d.ctx.fronts = fr; d.ctx.rowidx = ri; d.ctx.rel = re; d.ctx.children = ch;
    { i64 *p; UP(p, S.gth_ptr); d.ctx.gth_ptr = p; }
    { i64 *p; UP(p, S.gth_src); d.ctx.gth_src = p; }
    UP(d.fa_tasks, S.fa_tasks);
    UP(d.ea_tasks, S.ea_tasks); UP(d.potrf_tasks, S.potrf_tasks); UP(d.trsm_tasks, S.trsm_tasks);
    UP(d.update_tasks, S.update_tasks); UP(d.reduce_tasks, S.reduce_tasks);
    UP(d.chain_items, S.chain_items); d.n_chain_cnt = S.chain_counters;
    if (S.chain_counters > 0 && (rc = dev_alloc(h, &d.chain_cnt, S.chain_counters)) != TLPK_OK) return rc;
    if (!S.chain_items.empty() && std::getenv("TLPK_CHAIN_TRACE") && (rc = dev_alloc(h, &d.chain_trace, 4 * (i64)S.chain_items.size())) != TLPK_OK) return rc;
    { i32 *p; UP(p, S.upd_seg); d.ctx.upd_seg = p; }
    d.n_single = (i64)S.single_col.size();
    UP(d.single_loff, S.single_loff); UP(d.single_dinvoff, S.single_dinvoff); UP(d.single_col, S.single_col);
    UP(d.zero_tasks, S.zero_tasks); d.n_zero_tasks = (i64)S.zero_tasks.size() / 2; d.n_zero_lower = d.has_upper ? S.n_zero_lower : d.n_zero_tasks;
    UP(d.zero_small, S.zero_small); d.n_zero_small = (i64)S.zero_small.size();
    UP(d.fwd_gather_tasks, S.fwd_gather_tasks); UP(d.fwd_diag_tasks, S.fwd_diag_tasks);
    UP(d.fwd_update_tasks, S.fwd_update_tasks); UP(d.bwd_update_tasks, S.bwd_update_tasks);
    UP(d.fwd_small_tasks, S.fwd_small_tasks); UP(d.bwd_small_tasks, S.bwd_small_tasks);
    UP(d.fwd_sweep_tasks, S.fwd_sweep_tasks); UP(d.bwd_sweep_tasks, S.bwd_sweep_tasks);
#undef UP
#define AL(dst, cnt) if ((rc = dev_alloc(h, &(dst), (cnt))) != TLPK_OK) return rc
    AL(d.ctx.Lval, S.lval_len); AL(d.ctx.U0, S.ubuf_len[0]); AL(d.ctx.U1, S.ubuf_len[1]);
    // debugging aid: start from NaNs everywhere, so that a read of storage the factorisation never writes would show
    if (std::getenv("TLPK_POISON") && S.lval_len > 0) { HIPCHK(h, hipMemset(d.ctx.Lval, 0xFF, (size_t)S.lval_len * 8)); HIPCHK(h, hipDeviceSynchronize()); }
    AL(d.ctx.uc, 2 * S.uc_len); AL(d.ctx.xw, 2 * S.m); AL(d.ctx.info, 16);      // two copies: the second right-hand side of tlpk_solve2_device
    d.ctx.xw2 = S.m; d.ctx.uc2 = S.uc_len; AL(d.ctx.dinv, S.dinv_len); AL(d.ctx.spart, S.spart_len);
    const i64 nn = std::max<i64>(S.n, S.k2_n + 1);             // K2: user vectors have k2_n entries, D2 one more
    AL(h->d_theta, nn); AL(h->d_regP, nn); AL(h->d_regD, S.m); AL(h->d_D, nn);
    AL(h->d_xip, S.m); AL(h->d_xid, nn); AL(h->d_dx, nn); AL(h->d_dy, S.m);
    AL(d.rhs_w, std::max<i64>(2 * S.n, 1));
    d.asm_D = h->d_D; d.asm_regD = h->d_regD;
    // a shard of a multi-device handle receives only its slices of the input vectors: the rest stays zero (never used in arithmetic
    // that reaches a result, but never uninitialised either)
    HIPCHK(h, hipMemset(h->d_theta, 0, (size_t)nn * 8)); HIPCHK(h, hipMemset(h->d_regP, 0, (size_t)nn * 8)); HIPCHK(h, hipMemset(h->d_xid, 0, (size_t)nn * 8));
    HIPCHK(h, hipMemset(h->d_regD, 0, (size_t)std::max<i64>(S.m, 1) * 8)); HIPCHK(h, hipMemset(h->d_xip, 0, (size_t)std::max<i64>(S.m, 1) * 8));
    if (h->refine_steps > 0) { AL(h->d_r1, S.m); AL(h->d_r2, nn); AL(h->d_cx, nn); AL(h->d_cy, S.m); AL(h->d_ref, 8); }
    d.ctx.csign = nullptr;
    d.ctx.small_full = std::getenv("TLPK_SMALL_FULL") ? std::atoi(std::getenv("TLPK_SMALL_FULL")) : 0;
    d.ctx.upd_remap = 2;
    if (const char *e = std::getenv("TLPK_UPD_REMAP")) d.ctx.upd_remap = std::atoi(e);      // tuning knob
    if (S.system == 1) { double *p; if ((rc = dev_upload(h, &p, S.csign)) != TLPK_OK) return rc; d.ctx.csign = p; }
#undef AL
    {
        // persistent sweeps: one ticket counter per sweep launch, one hand-over word per column and direction
        i32 nslots = 0;
        h->sweep_slot_fwd.assign(S.fwd_launches.size(), -1); h->sweep_slot_bwd.assign(S.bwd_launches.size(), -1);
        for (size_t i = 0; i < S.fwd_launches.size(); ++i) if (S.fwd_launches[i].kind == LK_FWD_SWEEP) h->sweep_slot_fwd[i] = nslots++;
        for (size_t i = 0; i < S.bwd_launches.size(); ++i) if (S.bwd_launches[i].kind == LK_BWD_SWEEP) h->sweep_slot_bwd[i] = nslots++;
        // tickets and hand-over words in ONE allocation (tickets first, padded to 16 words): one memset per solve resets both
        const i64 nt = ((i64)nslots + 15) / 16 * 16;
        unsigned long long *blk = nullptr;
        if ((rc = dev_alloc(h, &blk, nt + 4 * S.m)) != TLPK_OK) return rc;       // [forward | backward] x two right-hand sides
        d.sweep_tickets = blk; d.sweep_xh = reinterpret_cast<double *>(blk + nt);
        d.sweep_reset_bytes = (nt + 2 * S.m) * 8; d.sweep_reset_bytes2 = (nt + 4 * S.m) * 8;
        HIPCHK(h, hipMemset(d.ctx.info, 0, 16 * sizeof(int)));
    }
    HIPCHK(h, hipHostMalloc((void **)&h->h_info, 4 * sizeof(int), hipHostMallocDefault));
    h->h_info[0] = h->h_info[1] = h->h_info[2] = h->h_info[3] = 0;
    // free host-side copies that are only needed on the device
    uvec<double>().swap(S.pair_w); uvec<i32>().swap(S.pair_j);
    return TLPK_OK;
}

void find_markers(tlpk_handle *h) {
    h->factor_marker = h->S.factor_launches.size();
    h->fwd_marker = h->S.fwd_launches.size();
    for (size_t i = 0; i < h->S.factor_launches.size(); ++i)
        if (h->S.factor_launches[i].kind == LK_ALLREDUCE_ROOT) h->factor_marker = i;
    for (size_t i = 0; i < h->S.fwd_launches.size(); ++i)
        if (h->S.fwd_launches[i].kind == LK_ALLREDUCE_ROOT) h->fwd_marker = i;
}

// ---- hipGraph replay of the static schedules -------------------------------------------------------------------------
// The launch schedules never change over the life of a handle: `update!` is ~220 launches on up to four streams, a `solve!`
// ~40.  The first call with a given set of argument pointers records the enqueue sequence with stream capture (cross-stream
// fork / join events become graph edges), later calls replay the instantiated graph with one hipGraphLaunch.  What it buys is
// host launch time and inter-kernel gaps, i.e. the small, latency-bound LPs (profiles/r03_small_lp_graph.txt); TLPK_GRAPH=0
// turns it off, profile mode and the split-phase (sharded) calls never use it.  The sweep kernels need no per-launch host
// state (their tickets are reset by the solve's own memset), every pointer in a captured node is either handle-owned or part
// of the cache key.
struct GraphKey { int kind; const void *p[8]; bool operator==(const GraphKey &o) const { return kind == o.kind && std::memcmp(p, o.p, sizeof(p)) == 0; } };

// Graphs are used for schedules with ONE stream group (general sparse LPs -- the small, launch-bound ones are of this kind; their
// side-stream fork / join is captured).  Block-angular LPs run two concurrent stream groups + side streams: capturing those four
// streams crashed (SIGSEGV inside the HIP runtime) in processes that had loaded PyTorch's bundled runtime first, while the same
// capture passes on the system runtime; the launches of these large LPs are hidden behind the kernels anyway
// (profiles/r03_small_lp_graph.txt).  TLPK_GRAPH=2 forces graphs for every schedule, TLPK_GRAPH=0 turns them off.
inline bool graph_usable(const tlpk_handle *h) {
    return h->use_graph && !h->profile && !h->serial && (h->S.ngroups <= 1 || h->force_graph);
}

// Solves: since round 5 the solve schedule of a block-angular LP is ONE schedule on the main stream (symbolic.cpp: solve_one_group), so a solve can be
// captured whatever the number of stream groups the factorisation uses.  What it buys there is host time: the blocking host-pointer solve
// (tlpk_solve) starts with ~25 short launches.  MEASURED (profiles/r05_host_path.txt) and OFF by default: config C4 52.85 vs 52.66 ms per step, north-star LP
// 139.0 vs 139.9, host-pointer path 157.3 vs 155.4 -- the host enqueues a solve in 20 - 70 us, there is nothing to hide.  TLPK_GRAPH_SOLVE=1 turns it on; never
// while an asynchronous update's root front is pending (the solve then waits for an event recorded outside the capture).
inline bool graph_usable_solve(const tlpk_handle *h) {
    static const bool on = [] { const char *e = std::getenv("TLPK_GRAPH_SOLVE"); return e && std::atoi(e) != 0; }();
    if (graph_usable(h)) return true;
    return on && h->use_graph && !h->profile && !h->serial && h->S.solve_single_stream && !h->root_pending && h->opt.nranks == 1;
}

template <class F>
int graph_or_direct(tlpk_handle *h, const GraphKey &key, F &&body, bool usable) {
    if (!usable) return body();
    for (size_t i = 0; i < h->graph_keys.size(); ++i)
        if (*reinterpret_cast<const GraphKey *>(h->graph_keys[i].data()) == key) {
            HIPCHK(h, hipGraphLaunch(h->graph_execs[i], h->stream));
            return TLPK_OK;
        }
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); h->use_graph = false; return body(); }
    const int rc = body();
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipStreamEndCapture(h->stream, &g);
    if (e == hipSuccess && rc == TLPK_OK) e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    if (g) hipGraphDestroy(g);
    if (e != hipSuccess || rc != TLPK_OK || !exec) {
        // capture is not available here (or the body failed): nothing has run yet -- enqueue directly from now on
        (void)hipGetLastError();
        h->use_graph = false;
        if (exec) hipGraphExecDestroy(exec);
        return rc != TLPK_OK ? rc : body();
    }
    if (h->graph_execs.size() >= 24) {                  // bounded cache (the interior-point loops use a handful of pointer sets)
        (void)hipStreamSynchronize(h->stream);          // the evicted graph may still be in flight
        hipGraphExecDestroy(h->graph_execs.front());
        h->graph_execs.erase(h->graph_execs.begin()); h->graph_keys.erase(h->graph_keys.begin());
    }
    h->graph_execs.push_back(exec);
    h->graph_keys.emplace_back(reinterpret_cast<const char *>(&key), reinterpret_cast<const char *>(&key) + sizeof(key));
    HIPCHK(h, hipGraphLaunch(exec, h->stream));
    return TLPK_OK;
}

// user-visible dimensions: for K2 the Symbolic describes the augmented matrix (order n + m)
inline i64 user_n(const tlpk_handle *h) { return h->S.system == 1 ? h->S.k2_n : h->S.n; }
inline i64 user_m(const tlpk_handle *h) { return h->S.system == 1 ? h->S.k2_m : h->S.m; }

}  // namespace

// diagnostic of the last FAILED tlpk_create / tlpk_create_multi of this thread: a failed create returns no handle, so tlpk_last_error
// has nothing to be asked on (round-3 advisor finding: the messages "no block-angular structure found", the RCCL / shard memory-gate texts
// could never reach the caller)
static thread_local std::string g_create_error;

extern "C" {

const char *tlpk_last_create_error(void) { return g_create_error.c_str(); }

void tlpk_default_options(tlpk_options *opt) {
    if (!opt) return;
    std::memset(opt, 0, sizeof(*opt));
    opt->struct_size = (int32_t)sizeof(tlpk_options);
    opt->device = 0;
    opt->ordering = TLPK_ORDER_AMD;
    opt->relax = 1;
    opt->nranks = 1;
}

int tlpk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tlpk_host_copy_threads(void) { return host_copy_threads(); }

// tlpk_create in two steps (tlpk_create_multi runs the first one once for all shards and the second per device):
//   create_host   : options -> handle, block detection, host analyse (or: copy of an analysed Symbolic + this rank's part)
//   create_device : streams / events, memory gate, upload
static int create_host(tlpk_handle *h, const tlpk_options &def, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval,
                       const double *nzval, int index_base, const Symbolic *common) {
    int rc = TLPK_OK;
    h->opt.ordering = def.ordering; h->opt.relax = def.relax;
    h->opt.rank = def.rank; h->opt.nranks = def.nranks < 1 ? 1 : def.nranks;
    h->opt.streams = def.streams;
    h->opt.system = (def.system == TLPK_SYSTEM_K2) ? 1 : 0;
    h->refine_steps = def.refine_steps;
    if (def.refine_steps < 0 || (def.refine_steps > 0 && (h->opt.system == 1 || h->opt.nranks > 1))) {
        h->last_error = "refine_steps: K1 on one rank only, >= 0";
        rc = TLPK_BADARG;
    }
    if (def.row_block && m > 0) {
        h->row_block_copy.assign(def.row_block, def.row_block + m);
        h->opt.row_block = h->row_block_copy.data();
    } else if (def.detect_blocks && m > 1 && colptr && !common) {
        // the hook that survives Tulip's presolve: find the block structure of the matrix KKT.setup actually received
        h->row_block_copy.assign((size_t)m, 0);
        int64_t nb = 1, nl = 0;
        const int drc = tlpk_detect_blocks(m, n, colptr, rowval, index_base, def.max_link_rows, h->row_block_copy.data(), &nb, &nl);
        if (drc != TLPK_OK && drc != TLPK_BADARG) { rc = drc; h->last_error = "tlpk_detect_blocks failed"; }
        if (drc == TLPK_OK && nb >= 2) h->opt.row_block = h->row_block_copy.data();
        else h->row_block_copy.clear();          // no structure (or malformed input: analyse reports it): general sparse path
    }
    if (def.ordering == TLPK_ORDER_USER && def.user_perm && m > 0) {
        h->user_perm_copy.resize((size_t)m);
        for (i64 i = 0; i < m; ++i) h->user_perm_copy[(size_t)i] = def.user_perm[i] - index_base;
        h->opt.user_perm = h->user_perm_copy.data();
    }
    h->profile = def.profile != 0;
    if (const char *e = std::getenv("TLPK_SERIAL")) h->serial = std::atoi(e) != 0;
    if (const char *e = std::getenv("TLPK_GRAPH")) { h->use_graph = std::atoi(e) != 0; h->force_graph = std::atoi(e) >= 2; }
    if (const char *e = std::getenv("TLPK_POLL")) std::sscanf(e, "%d,%d,%d", &h->poll[0], &h->poll[1], &h->poll[2]);
    if (const char *e = std::getenv("TLPK_STAGGER")) { h->stagger = std::atoi(e) != 0; if (std::atoi(e) > 1) h->stagger_min = std::atoi(e); }
    if (const char *e = std::getenv("TLPK_CHAIN_FAULT")) h->fault_at = std::atoi(e);
    h->opt.shared_device = h->shared_device ? 1 : 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (rc == TLPK_OK) {
        if (common) { h->S = *common; h->opt.k2_n = common->k2_n; rc = analyse_rank(h->S, h->opt); }
        else rc = (h->opt.system == 1) ? analyse_k2(h->S, m, n, colptr, rowval, nzval, index_base, h->opt)
                                       : analyse(h->S, m, n, colptr, rowval, nzval, index_base, h->opt);
    }
    h->ms_analyse = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->last_error = h->S.error.empty() ? h->last_error : h->S.error;
    if (rc == TLPK_OK) {
        find_markers(h);
        h->nlink = (h->S.root_front >= 0) ? h->S.fronts[h->S.root_front].ns : 0;
        h->first_link = h->S.m - h->nlink;
    }
    return rc;
}

static int create_device(tlpk_handle *h, const tlpk_options &def) {
    int rc = TLPK_OK;
    if (def.device >= 0) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || def.device >= ndev) {
            h->last_error = "no HIP device " + std::to_string(def.device) + " visible";
            return TLPK_NO_DEVICE;
        }
        h->device = def.device;
        hipError_t e = hipSetDevice(h->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        h->gstream[0] = h->stream;
        // Only the streams the schedule uses, created back to back: the runtime multiplexes
        // streams onto a few hardware queues (4 by default) in creation order, and two of OUR
        // streams on one queue serialise (measured: 70 vs 80 ms/step on C4 depending on what
        // else the process had created before).
        const int ng = std::max(1, h->S.ngroups);
        for (int g = 1; g < ng && e == hipSuccess; ++g) e = hipStreamCreateWithFlags(&h->gstream[g], hipStreamNonBlocking);
        for (int g = 0; g < ng && e == hipSuccess; ++g) e = hipStreamCreateWithFlags(&h->sstream[g], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
        if (e == hipSuccess && ng >= 2) {
            // lowest priority: slots go to the leaf levels' launches first.  (A CU-masked stream -- 32 / 64 / 128 of the 256 CUs, to leave HBM bandwidth
            // to the leaf levels -- measured no better: profiles/r04_defer_upper.txt.)
            int least = 0, greatest = 0;
            hipDeviceGetStreamPriorityRange(&least, &greatest);
            e = hipStreamCreateWithPriority(&h->zstream, hipStreamNonBlocking, least);
        }
        if (e == hipSuccess && ng >= 2) e = hipEventCreateWithFlags(&h->ev_zfork, hipEventDisableTiming);
        if (e == hipSuccess && ng >= 2) e = hipEventCreateWithFlags(&h->ev_upper, hipEventDisableTiming);
        for (int g = 1; g < ng && e == hipSuccess; ++g) e = hipEventCreateWithFlags(&h->ev_join[g], hipEventDisableTiming);
        for (int g = 0; g < ng && e == hipSuccess; ++g) e = hipEventCreateWithFlags(&h->ev_side[g], hipEventDisableTiming);
        if (e == hipSuccess && h->stagger) e = hipEventCreateWithFlags(&h->ev_stagger, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&h->ev0);
        if (e == hipSuccess) e = hipEventCreate(&h->ev1);
        if (e == hipSuccess && h->S.root_front >= 0 && h->opt.nranks == 1) {       // a stream of its own for the root front (tlpk_update_device_async)
            e = hipStreamCreateWithFlags(&h->rstream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_blocks, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_root, hipEventDisableTiming);
        }
        if (e != hipSuccess) rc = hip_fail(h, e, "device init");
        if (rc == TLPK_OK) {
            // memory gate: refuse before allocating
            size_t free_b = 0, total_b = 0;
            hipMemGetInfo(&free_b, &total_b);
            const double budget = def.mem_budget_bytes > 0 ? (double)def.mem_budget_bytes : 0.9 * (double)free_b;
            const double need = 8.0 * ((double)h->S.lval_len + (double)h->S.ubuf_len[0] + (double)h->S.ubuf_len[1] +
                                       (double)h->S.spart_len + (double)h->S.dinv_len + (double)h->S.uc_len +
                                       (double)h->S.gth_ptr.size() + (double)h->S.gth_src.size()) +
                                12.0 * (double)h->S.pair_w.size() + 21.0 * (double)h->S.nnzS + 52.0 * (double)h->S.nnzA +
                                8.0 * (double)h->S.rowidx.size() +
                                (double)sizeof(UpdateTask) * (double)(h->S.update_tasks.size() + h->S.reduce_tasks.size()) +
                                (double)sizeof(EaTask) * (double)h->S.ea_tasks.size() + (double)sizeof(TrsmTask) * (double)h->S.trsm_tasks.size() +
                                4.0 * (double)h->S.upd_seg.size() + 32.0 * (double)h->S.m;
            if (need > budget) {
                h->last_error = "factor needs " + std::to_string(need / 1e9) + " GB, budget " + std::to_string(budget / 1e9) + " GB";
                if (h->S.system == 0 && !h->S.Ap.empty()) {
                    // K1 forms A D A': one column of A with c entries makes a c x c dense block of S ("dense columns")
                    i64 cmax = 0, jmax = -1;
                    for (i64 j = 0; j < h->S.n; ++j) { const i64 cj = h->S.Ap[(size_t)j + 1] - h->S.Ap[(size_t)j]; if (cj > cmax) { cmax = cj; jmax = j; } }
                    if (cmax >= 1000 && (double)cmax * (double)cmax >= 0.05 * (double)h->S.nnzL)
                        h->last_error += "; column " + std::to_string(jmax) + " of A has " + std::to_string(cmax) + " entries (a dense column fills the normal equations): "
                                         "the augmented system KKT_System = K2 (TLPK_SYSTEM_K2) does not form A*D*A'";
                }
                rc = TLPK_TOO_LARGE;
            }
        }
        if (rc == TLPK_OK) rc = upload_all(h);
        if (rc == TLPK_OK) h->has_device = true;
    } else if (def.mem_budget_bytes > 0) {
        const double need = 8.0 * ((double)h->S.lval_len + (double)h->S.ubuf_len[0] + (double)h->S.ubuf_len[1]);
        if (need > (double)def.mem_budget_bytes) { h->last_error = "factor exceeds mem_budget_bytes"; rc = TLPK_TOO_LARGE; }
    }
    return rc;
}

int tlpk_create(tlpk_handle **out, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval,
                const double *nzval, int index_base, const tlpk_options *uopt) {
    if (!out) return TLPK_BADARG;
    *out = nullptr;
    tlpk_options def;
    tlpk_default_options(&def);
    if (uopt) {
        if (uopt->struct_size != (int32_t)sizeof(tlpk_options)) return TLPK_BADARG;
        def = *uopt;
    }
    tlpk_handle *h = new (std::nothrow) tlpk_handle();
    if (!h) return TLPK_OOM;
    int rc = TLPK_OK;
    try {
        rc = create_host(h, def, m, n, colptr, rowval, nzval, index_base, nullptr);
        if (rc == TLPK_OK) rc = create_device(h, def);
    } catch (const std::bad_alloc &) {
        rc = TLPK_OOM; h->last_error = "host out of memory during analyse";
    } catch (...) {
        rc = TLPK_INTERNAL; h->last_error = "unexpected exception";
    }
    if (rc != TLPK_OK) {
        g_create_error = h->last_error;
        // no live handle on failure (a C caller that treats rc != 0 as "no handle" would leak the host symbolic data) -- unless the caller
        // asks for the analyse-only handle that describes what did not fit
        if (!(rc == TLPK_TOO_LARGE && def.keep_on_too_large)) { tlpk_destroy(h); return rc; }
    } else g_create_error.clear();
    *out = h;
    return rc;
}

namespace { void multi_comm_destroy(void *comm); }

void tlpk_destroy(tlpk_handle *h) {
    if (!h) return;
    if (!h->sub.empty() || h->multi_tmp) {                // multi-device parent: owns its per-device handles, nothing else
        if (h->multi_rccl) for (size_t r = 0; r < h->sub.size(); ++r) if (h->multi_comm[r]) multi_comm_destroy(h->multi_comm[r]);
        shard_pool_delete(h->shard_pool); h->shard_pool = nullptr;
        for (tlpk_handle *c : h->sub) tlpk_destroy(c);
        ipm_free(h);
        if (h->multi_tmp) { hipSetDevice(h->device); hipFree(h->multi_tmp); }
        if (h->multi_done) hipEventDestroy(h->multi_done);
        for (int r = 0; r < MAX_DEVICES; ++r) { if (h->multi_ev[r]) hipEventDestroy(h->multi_ev[r]); if (h->multi_ev2[r]) hipEventDestroy(h->multi_ev2[r]); }
        delete h;
        return;
    }
    if (h->device >= 0) {
        hipSetDevice(h->device);
        if (h->stream) hipStreamSynchronize(h->stream);
        ipm_free(h);
        for (hipGraphExec_t g : h->graph_execs) hipGraphExecDestroy(g);
        for (void *p : h->allocs) hipFree(p);
        if (h->h_info) hipHostFree(h->h_info);
        if (h->pin_in) hipHostFree(h->pin_in);
        if (h->pin_out) hipHostFree(h->pin_out);
        for (hipEvent_t e : h->io_events) hipEventDestroy(e);
        for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
        if (h->ev0) hipEventDestroy(h->ev0);
        if (h->ev1) hipEventDestroy(h->ev1);
        if (h->ev_fork) hipEventDestroy(h->ev_fork);
        if (h->zstream) { hipStreamSynchronize(h->zstream); hipStreamDestroy(h->zstream); }
        if (h->ev_zfork) hipEventDestroy(h->ev_zfork);
        if (h->ev_upper) hipEventDestroy(h->ev_upper);
        if (h->rstream) { hipStreamSynchronize(h->rstream); hipStreamDestroy(h->rstream); }
        if (h->ev_blocks) hipEventDestroy(h->ev_blocks);
        if (h->ev_root) hipEventDestroy(h->ev_root);
        if (h->ev_stagger) hipEventDestroy(h->ev_stagger);
        for (int g = 0; g < MAX_GROUPS; ++g) {
            if (h->ev_side[g]) hipEventDestroy(h->ev_side[g]);
            if (h->sstream[g]) { hipStreamSynchronize(h->sstream[g]); hipStreamDestroy(h->sstream[g]); }
        }
        for (int g = 1; g < MAX_GROUPS; ++g) {
            if (h->ev_join[g]) hipEventDestroy(h->ev_join[g]);
            if (h->gstream[g]) { hipStreamSynchronize(h->gstream[g]); hipStreamDestroy(h->gstream[g]); }
        }
        if (h->stream) hipStreamDestroy(h->stream);
    }
    delete h;
}

// ---- update ----
static int update_async_wait(tlpk_handle *h);
// everything of an update up to the reduction of the root panel, on the handle-owned copies of theta / regP / regD
static int enq_update_local(tlpk_handle *h) {
    const Symbolic &S = h->S;
    HIPCHK(h, hipMemcpyAsync(h->d.ctx.info, h->h_info, sizeof(int), hipMemcpyHostToDevice, h->stream));
    // words 1..15: the "somebody gave up waiting" flag of the dependency-driven launches / the sweeps and its diagnostics.  They were cleared at set-up only: after ONE
    // launch that gave up every later update of the handle saw the flag, skipped its work and came back with TLPK_INTERNAL within milliseconds (found at the end of round 6
    // by 2 500 updates in a row on eight shards of one GPU: 2 448 failures behind the first; a replay of the update could not succeed either)
    HIPCHK(h, hipMemsetAsync(h->d.ctx.info + 1, 0, 15 * sizeof(int), h->stream));
    // TLPK_CHAIN_FAULT=k (testing): update number k of the handle (0-based, first attempt only) starts with the flag SET, as if a workgroup had given up waiting: every
    // dependency-driven launch of it skips its work and the update ends TLPK_INTERNAL -- what tlpk_update's replay and the next update must recover from
    if (h->d.n_chain_cnt > 0 && h->fault_at >= 0 && h->n_updates == h->fault_at && !h->fault_done) {
        h->fault_done = true;
        HIPCHK(h, hipMemsetAsync(h->d.ctx.info + 1, 1, sizeof(int), h->stream));
    }
    ++h->n_updates;
    if (h->d.n_chain_cnt > 0) HIPCHK(h, hipMemsetAsync(h->d.chain_cnt, 0, (size_t)h->d.n_chain_cnt * sizeof(unsigned), h->stream));     // tickets + completion counters of the chain launches
    {
        ProfScope ps(h, TLPK_KC_ASSEMBLE);
        if (S.system == 1) launch_k2_diag(h->stream, user_n(h), h->d_theta, h->d_regP, h->d_D);      // D2 = [theta + regP ; 1]  (sqd.jl:44-50)
        else launch_compute_d(h->stream, S.n, h->d_theta, h->d_regP, h->d_D);
    }
    // step 13d: zero-fill + assembly of the upper fronts (97 % of the factor's bytes on a block-angular LP) on the last stream group's side stream,
    // beside the latency-bound leaf levels; the groups wait for ev_upper at their LK_WAIT_UPPER marker.  One-stream order in the single-stream modes.
    h->upper_split = h->d.has_upper && h->zstream && h->ev_upper && S.ngroups >= 2 && !h->profile && !h->serial && !graph_usable(h);
    if (h->upper_split) {
        hipStream_t zs = h->zstream;
        HIPCHK(h, hipEventRecord(h->ev_zfork, h->stream));                 // D is computed, the previous factor is no longer read
        launch_zero_panels(h->stream, h->d, 0);                            // the lower fronts first: their (short) workgroups take their slots before the long fill starts
        launch_assemble(h->stream, h->d, h->d_D, h->d_regD, 0);
        launch_single_factor(h->stream, h->d);
        HIPCHK(h, hipStreamWaitEvent(zs, h->ev_zfork, 0));
        launch_zero_panels(zs, h->d, 1);
        launch_assemble(zs, h->d, h->d_D, h->d_regD, 1);
        HIPCHK(h, hipEventRecord(h->ev_upper, zs));
    } else {
        ProfScope ps(h, TLPK_KC_ASSEMBLE);
        launch_zero_panels(h->stream, h->d);
        launch_assemble(h->stream, h->d, h->d_D, h->d_regD);
        launch_single_factor(h->stream, h->d);
    }
    run_launches(h, S.factor_launches, 0, h->factor_marker);
    HIPCHK(h, hipGetLastError());
    return TLPK_OK;
}
// the root front and the read-back of the status word
static int enq_update_finish(tlpk_handle *h, hipStream_t root_stream = nullptr) {
    const Symbolic &S = h->S;
    run_launches(h, S.factor_launches, h->factor_marker, S.factor_launches.size(), -1, 1, root_stream);
    // (with chain launches also info[1]: a workgroup of k_chain gave up waiting -- the bounded spins of kernels.hip)
    HIPCHK(h, hipMemcpyAsync(h->h_info, h->d.ctx.info, (h->d.n_chain_cnt > 0 ? 2 : 1) * sizeof(int), hipMemcpyDeviceToHost, root_stream ? root_stream : h->stream));
    return TLPK_OK;
}

int tlpk_update_local(tlpk_handle *h, const double *d_theta, const double *d_regP, const double *d_regD) {
    if (!h || !d_theta || !d_regP || !d_regD) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    HIPCHK(h, hipSetDevice(h->device));
    const Symbolic &S = h->S;
    (void)S;
    if (h->root_pending) { (void)update_async_wait(h); }      // an unchecked tlpk_update_device_async: complete it (its status is the caller's loss)
    h->factored = false; h->local_done = false; h->solve_local_done = false; h->fail_col = -1; h->solve_timed = false;
    prof_begin(h, true);
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    // stored copies (spd.jl:36-38): the caller may mutate its vectors right after the call
    const i64 un = user_n(h), um = user_m(h);
    if (d_theta != h->d_theta) HIPCHK(h, hipMemcpyAsync(h->d_theta, d_theta, (size_t)un * 8, hipMemcpyDeviceToDevice, h->stream));
    if (d_regP != h->d_regP) HIPCHK(h, hipMemcpyAsync(h->d_regP, d_regP, (size_t)un * 8, hipMemcpyDeviceToDevice, h->stream));
    if (d_regD != h->d_regD) HIPCHK(h, hipMemcpyAsync(h->d_regD, d_regD, (size_t)um * 8, hipMemcpyDeviceToDevice, h->stream));
    h->h_info[0] = INT_MAX;
    if (h->update_whole) return TLPK_OK;                 // tlpk_update_device on an unsharded handle: the caller enqueues both halves (graph)
    if (int rc = enq_update_local(h)) return rc;
    h->local_done = true;
    return TLPK_OK;
}

int tlpk_root_panel(tlpk_handle *h, double **d_ptr, int64_t *count) {
    if (!h || !d_ptr || !count) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (h->S.root_front < 0) { *d_ptr = nullptr; *count = 0; return TLPK_OK; }
    const FrontDesc &fd = h->S.fronts[h->S.root_front];
    *d_ptr = h->d.ctx.Lval + fd.loff;
    *count = pk_len(fd.lda, fd.ns);          // whole (packed) panel incl. the alignment rows (they stay zero)
    return TLPK_OK;
}

int tlpk_root_copy(tlpk_handle *h, int which, int dir, double *d_buf) {
    if (!h || !d_buf || which < 0 || which > 2 || (dir != 0 && dir != 1)) return TLPK_BADARG;
    if (!h->has_device) return TLPK_NO_DEVICE;
    double *p = nullptr; int64_t n = 0;
    int rc = which == 0 ? tlpk_root_panel(h, &p, &n) : (which == 1 ? tlpk_root_rhs(h, &p, &n) : tlpk_root_rhs2(h, &p, &n));
    if (rc != TLPK_OK || n == 0) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(dir == 0 ? d_buf : p, dir == 0 ? p : d_buf, (size_t)n * 8, hipMemcpyDeviceToDevice, h->stream));
    return TLPK_OK;
}

// tlpk_update_finish = enqueue (root front + status read-back) + wait.  The multi-device mode enqueues the finish of EVERY shard
// before it waits for any of them: with the synchronisation inside, shard r's root factorisation was not even enqueued before
// shard r - 1's had completed (N - 1 serialised root fronts per update).
static int update_finish_enqueue(tlpk_handle *h) {
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (!h->local_done) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = enq_update_finish(h)) return rc;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    return TLPK_OK;
}
static int update_finish_wait(tlpk_handle *h) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    float ms = 0.f; hipEventElapsedTime(&ms, h->ev0, h->ev1); h->ms_update = ms;
    prof_collect(h);
    h->local_done = false;
    if (h->d.n_chain_cnt > 0 && h->h_info[1] != 0) {
        if (std::getenv("TLPK_CHAIN_DEBUG")) {
            int dbg[16] = {0}; (void)hipMemcpy(dbg, h->d.ctx.info, sizeof(dbg), hipMemcpyDeviceToHost);
            std::fprintf(stderr, "[tlpk chain] rank %d: gave up at item %d (role %d, task %d): counter %d needs %d, has %d; workgroup %d of %d\n", h->opt.rank, dbg[4], dbg[5], dbg[11], dbg[6], dbg[7], dbg[8], dbg[9], dbg[10]);
            std::vector<unsigned> cnt((size_t)h->d.n_chain_cnt); (void)hipMemcpy(cnt.data(), h->d.chain_cnt, cnt.size() * 4, hipMemcpyDeviceToHost);
            std::fprintf(stderr, "[tlpk chain] counters:");
            for (size_t q = 0; q < std::min<size_t>(cnt.size(), 12); ++q) std::fprintf(stderr, " %u", cnt[q]);
            std::fprintf(stderr, " ... (%zu words)\n", cnt.size());
            for (const Launch &L : h->S.factor_launches) if (L.kind == LK_CHAIN) std::fprintf(stderr, "[tlpk chain] launch: %lld items, ticket word %d = %u\n", (long long)L.count, L.pad, cnt[(size_t)L.pad]);
            if (h->d.chain_trace) {                               // TLPK_CHAIN_TRACE=1 as well: the items' time stamps go to the file TLPK_CHAIN_DEBUG names (+ .rank<r>.bin)
                std::vector<unsigned long long> tr(4 * h->S.chain_items.size());
                (void)hipMemcpy(tr.data(), h->d.chain_trace, tr.size() * 8, hipMemcpyDeviceToHost);
                const std::string path = std::string(std::getenv("TLPK_CHAIN_DEBUG")) + ".rank" + std::to_string(h->opt.rank) + ".bin";
                if (FILE *f = std::fopen(path.c_str(), "wb")) { std::fwrite(tr.data(), 8, tr.size(), f); std::fclose(f); std::fprintf(stderr, "[tlpk chain] time stamps of %zu items -> %s\n", h->S.chain_items.size(), path.c_str()); }
            }
        }
        h->last_error = "the dependency-driven factorisation gave up waiting for a completion counter (internal scheduling error); the factor is invalid"; return TLPK_INTERNAL; }
    if (h->h_info[0] != INT_MAX) { h->fail_col = h->h_info[0]; return TLPK_NOT_POSDEF; }
    h->factored = true;
    return TLPK_OK;
}

int tlpk_update_finish(tlpk_handle *h) {
    if (!h) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (int rc = update_finish_enqueue(h)) return rc;
    return update_finish_wait(h);
}

// The composed entry points run both halves back to back WITHOUT the all-reduce of the root panel /
// root rhs: on a sharded handle that would factorise this rank's partial sum and return TLPK_OK.
namespace {
int multi_update(tlpk_handle *h, const double *theta, const double *regP, const double *regD);
int multi_solve(tlpk_handle *h, double *dx, double *dy, const double *xi_p, const double *xi_d);
}
static int sharded_needs_split(tlpk_handle *h, const char *what) {
    if (h && h->opt.nranks > 1 && h->S.root_front >= 0) {
        h->last_error = std::string(what) + ": handle is sharded over " + std::to_string(h->opt.nranks) +
                        " ranks with a replicated root front; use the split-phase calls (tlpk_*_local, all-reduce, tlpk_*_finish)";
        return TLPK_BADARG;
    }
    return TLPK_OK;
}

// tlpk_update_device without the wait for its status word: everything is enqueued -- the root (linking) front on a stream of its own,
// so that the block-level forward sweeps of a following solve overlap its (nearly idle: one workgroup per 64 columns) factorisation --
// and the status is reported by the next tlpk_sync.  Solves enqueued in between are speculative: after a failed factorisation their
// results are meaningless and tlpk_sync returns TLPK_NOT_POSDEF.
static int update_async_wait(tlpk_handle *h) {
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->rstream) HIPCHK(h, hipStreamSynchronize(h->rstream));
    HIPCHK(h, hipGetLastError());
    float ms = 0.f; if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->ms_update = ms;
    h->root_pending = false; h->local_done = false;
    if (h->d.n_chain_cnt > 0 && h->h_info[1] != 0) { h->factored = false; h->last_error = "the dependency-driven factorisation gave up waiting for a completion counter (internal scheduling error); the factor is invalid"; return TLPK_INTERNAL; }
    if (h->h_info[0] != INT_MAX) { h->fail_col = h->h_info[0]; h->factored = false; return TLPK_NOT_POSDEF; }
    h->factored = true;
    return TLPK_OK;
}

int tlpk_update_device_async(tlpk_handle *h, const double *d_theta, const double *d_regP, const double *d_regD) {
    if (int g = sharded_needs_split(h, "tlpk_update_device_async")) return g;
    if (!h || !h->sub.empty() || !h->has_device || h->profile || h->serial || graph_usable(h) || !h->rstream)
        return tlpk_update_device(h, d_theta, d_regP, d_regD);          // nothing to overlap / a mode that serialises anyway: the blocking call
    h->update_whole = true;
    int rc = tlpk_update_local(h, d_theta, d_regP, d_regD);             // argument checks, stored copies, ev0
    h->update_whole = false;
    if (rc != TLPK_OK) return rc;
    if ((rc = enq_update_local(h)) != TLPK_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev_blocks, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->rstream, h->ev_blocks, 0));
    if ((rc = enq_update_finish(h, h->rstream)) != TLPK_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev1, h->rstream));
    HIPCHK(h, hipEventRecord(h->ev_root, h->rstream));
    HIPCHK(h, hipGetLastError());
    h->root_pending = true;
    h->factored = true;                                                  // tentatively: tlpk_sync delivers the verdict
    return TLPK_OK;
}

int tlpk_update_device(tlpk_handle *h, const double *d_theta, const double *d_regP, const double *d_regD) {
    if (int g = sharded_needs_split(h, "tlpk_update_device")) return g;
    if (!h || !h->sub.empty() || !h->has_device || !graph_usable(h)) {
        int rc = tlpk_update_local(h, d_theta, d_regP, d_regD);
        if (rc != TLPK_OK) return rc;
        return tlpk_update_finish(h);
    }
    // unsharded handle: prologue (stored copies of the caller's vectors), then the WHOLE factorisation as one replayed graph
    h->update_whole = true;
    int rc = tlpk_update_local(h, d_theta, d_regP, d_regD);
    h->update_whole = false;
    if (rc != TLPK_OK) return rc;
    const GraphKey key{1, {}};
    rc = graph_or_direct(h, key, [&]() { const int q = enq_update_local(h); return q != TLPK_OK ? q : enq_update_finish(h); }, graph_usable(h));
    if (rc != TLPK_OK) return rc;
    h->local_done = true;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    return update_finish_wait(h);
}

// Host-pointer entry points: the caller's vectors are ordinary pageable memory (Julia arrays, KKT.jl:83,100).  They go through pinned
// staging buffers owned by the handle, in pieces handled by a small pool of host threads (hostcopy.hpp): a piece is copied into the staging
// area and its hipMemcpyAsync issued by the same thread, so the link works from the first piece on (round 4: one thread, whole vectors,
// 13 - 19 GB/s end to end).  Nothing of the caller's is referenced after the call returns.
static int ensure_pinned(tlpk_handle *h) {
    if (h->pin_in) return TLPK_OK;
    const size_t nin = (size_t)std::max<i64>(2 * user_n(h) + user_m(h), 1), nout = (size_t)std::max<i64>(user_n(h) + user_m(h), 1);
    HIPCHK(h, hipHostMalloc((void **)&h->pin_in, nin * 8, hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void **)&h->pin_out, nout * 8, hipHostMallocDefault));
    return TLPK_OK;
}

namespace {
double now_ms();
struct HostVec { double *dev; double *host; i64 count; };        // one vector of a host-pointer call (host: source or destination)
// A vector crosses the link in DMA GROUPS of ~2 MB (one hipMemcpyAsync each; on the way out one event each) and is copied into / out of the staging
// area in CHUNKS of <= 512 KB by the pool's threads (several chunks per group).  First version of round 5: one 512 KB copy + event per piece -- the
// 15 device-to-host copies of a config-C4 solve took 0.4 ms for 7.7 MB (19 GB/s, gpurun_out session B: TLPK_HOSTIO_TIMING); the link wants few, large copies,
// the host copy wants many, small ones.  Also measured, and removed: the groups alternating between two extra copy streams (two copy engines) -- the device
// time of a solve rose from 2.45 to 3.3 ms with the two extra streams alive (more streams than hardware queues: the handle's main stream then shares one),
// host-pointer step 60.3 vs 56.2 ms (profiles/r05_host_path.txt).
struct IoGroup { double *dev, *pin; i64 cnt; std::atomic<int> left{0}; IoGroup() = default; IoGroup(const IoGroup &o) : dev(o.dev), pin(o.pin), cnt(o.cnt), left(o.left.load()) {} };
struct IoChunk { double *pin, *host; i64 cnt; int group; };
void io_plan(double *pin, const HostVec *v, int nv, std::vector<IoGroup> &groups, std::vector<IoChunk> &chunks) {
    i64 total = 0;
    for (int k = 0; k < nv; ++k) total += v[k].count;
    const i64 gsz = std::max<i64>(262144, ((total + 15) / 16 + 65535) / 65536 * 65536);      // >= 2 MB, at most ~16 groups per call
    const i64 csz = 65536;                                                                  // 512 KB
    groups.clear(); chunks.clear();
    i64 off = 0;
    for (int k = 0; k < nv; ++k) {
        for (i64 o = 0; o < v[k].count; o += gsz) {
            IoGroup g; g.dev = v[k].dev + o; g.pin = pin + off + o; g.cnt = std::min(gsz, v[k].count - o);
            int nch = 0;
            for (i64 c = 0; c < g.cnt; c += csz, ++nch) chunks.push_back(IoChunk{g.pin + c, v[k].host + o + c, std::min(csz, g.cnt - c), (int)groups.size()});
            g.left.store(nch);
            groups.push_back(g);
        }
        off += v[k].count;
    }
}
// host -> staging -> device, all vectors of a call; returns when every copy has been ENQUEUED on the handle's stream.  The thread that stages the
// last chunk of a group issues the group's copy: the link works while the other groups are still being staged.
int stage_in(tlpk_handle *h, const HostVec *v, int nv) {
    std::vector<IoGroup> groups; std::vector<IoChunk> chunks;
    io_plan(h->pin_in, v, nv, groups, chunks);
    std::atomic<int> err{(int)hipSuccess};
    const int dev = h->device; hipStream_t st = h->stream;
    host_parallel_for((int)chunks.size(), [&](int i) {
        const IoChunk &c = chunks[(size_t)i];
        copy_to_staging(c.pin, c.host, (size_t)c.cnt * 8);
        IoGroup &g = groups[(size_t)c.group];
        if (g.left.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
        hipError_t e = hipSetDevice(dev);                        // (per thread; a no-op after the first time)
        if (e == hipSuccess) e = hipMemcpyAsync(g.dev, g.pin, (size_t)g.cnt * 8, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) err.store((int)e);
    });
    if (err.load() != (int)hipSuccess) return hip_fail(h, (hipError_t)err.load(), "staged host-to-device copy");
    return TLPK_OK;
}
// device -> staging -> host: every group's copy is followed by an event; the pool copies a chunk out as soon as its group's event has fired,
// while the later groups are still on the link
int stage_out(tlpk_handle *h, const HostVec *v, int nv) {
    std::vector<IoGroup> groups; std::vector<IoChunk> chunks;
    io_plan(h->pin_out, v, nv, groups, chunks);
    while (h->io_events.size() < groups.size()) {
        hipEvent_t e = nullptr;
        HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->io_events.push_back(e);
    }
    for (size_t i = 0; i < groups.size(); ++i) {
        HIPCHK(h, hipMemcpyAsync(groups[i].pin, groups[i].dev, (size_t)groups[i].cnt * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipEventRecord(h->io_events[i], h->stream));
    }
    std::atomic<int> err{(int)hipSuccess};
    const int dev = h->device;
    h->io_t_first = h->io_t_last = 0.0;
    host_parallel_for((int)chunks.size(), [&](int i) {
        const IoChunk &c = chunks[(size_t)i];
        hipError_t e = hipSetDevice(dev);
        if (e == hipSuccess) e = hipEventSynchronize(h->io_events[(size_t)c.group]);
        if (e != hipSuccess) { err.store((int)e); return; }
        if (h->io_timing) { const double t = now_ms(); if (c.group == 0 && c.pin == groups[0].pin) h->io_t_first = t; if (c.group + 1 == (int)groups.size()) h->io_t_last = t; }
        std::memcpy(c.host, c.pin, (size_t)c.cnt * 8);
    });
    if (err.load() != (int)hipSuccess) return hip_fail(h, (hipError_t)err.load(), "staged device-to-host copy");
    return TLPK_OK;
}
// blocking parallel copy between two host buffers (multi-device mode: the job-wide vectors into / out of the lead's staging area)
void host_copy(double *dst, const double *src, i64 count, bool to_staging) {
    const i64 piece = std::max<i64>(65536, ((count + 31) / 32 + 8191) / 8192 * 8192);
    const int np = (int)((count + piece - 1) / piece);
    host_parallel_for(np, [&](int i) {
        const i64 o = (i64)i * piece, c = std::min(piece, count - o);
        if (to_staging) copy_to_staging(dst + o, src + o, (size_t)c * 8); else std::memcpy(dst + o, src + o, (size_t)c * 8);
    });
}
}  // namespace

static int update_once(tlpk_handle *h, const double *theta, const double *regP, const double *regD) {
    if (!h || !theta || !regP || !regD) return TLPK_BADARG;
    if (!h->sub.empty()) return multi_update(h, theta, regP, regD);
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (int g = sharded_needs_split(h, "tlpk_update")) return g;
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = ensure_pinned(h)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));          // the staging area may still feed an earlier call's copies
    const i64 un = user_n(h), um = user_m(h);
    const HostVec in[3] = {{h->d_theta, const_cast<double *>(theta), un}, {h->d_regP, const_cast<double *>(regP), un}, {h->d_regD, const_cast<double *>(regD), um}};
    static const bool timing = [] { const char *e = std::getenv("TLPK_HOSTIO_TIMING"); return e && std::atoi(e) != 0; }();
    const double t0 = timing ? now_ms() : 0.0;
    if (int rc = stage_in(h, in, 3)) return rc;
    const double t1 = timing ? now_ms() : 0.0;
    const int rc = tlpk_update_device(h, h->d_theta, h->d_regP, h->d_regD);
    if (timing) std::fprintf(stderr, "tlpk_update host path: stage in + H2D issued %.3f ms | factorisation (enqueue + wait) %.3f ms | device update (events) %.3f ms\n",
                             t1 - t0, now_ms() - t1, h->ms_update);
    return rc;
}

// A dependency-driven launch that gave up waiting (the bounded spins of k_chain: TLPK_INTERNAL, nothing hangs) is REPLAYED ONCE: the caller's vectors are unchanged,
// the counters start from zero with every update, the factorisation is deterministic.  Every such case seen so far was the stall of
// profiles/r06_chain_poll_storm.txt -- eight shards' launches on ONE GPU, a workgroup standing still inside its role while the rest of the device spins: 1 - 3 % of the
// runs of the eight-shards tests, never on a handle that has its device to itself --, not a schedule that cannot finish: a second failure is returned as it is.
// TLPK_CHAIN_RETRY=0: no replay; the number of replays of a handle: symbolic array "chain_retries".
static bool chain_gave_up(const tlpk_handle *h) {
    if (!h->sub.empty()) { for (const tlpk_handle *c : h->sub) if (chain_gave_up(c)) return true; return false; }
    return h->d.n_chain_cnt > 0 && h->h_info[1] != 0;
}
int tlpk_update(tlpk_handle *h, const double *theta, const double *regP, const double *regD) {
    int rc = update_once(h, theta, regP, regD);
    if (rc == TLPK_INTERNAL && h && chain_gave_up(h) && [] { const char *e = std::getenv("TLPK_CHAIN_RETRY"); return !e || std::atoi(e) != 0; }()) {
        ++h->chain_retries;
        if (std::getenv("TLPK_CHAIN_DEBUG")) std::fprintf(stderr, "[tlpk chain] a launch gave up waiting: the update is replayed once (replay %d of this handle)\n", h->chain_retries);
        rc = update_once(h, theta, regP, regD);
    }
    return rc;
}

// ---- solve ----
// rhs_rank >= 0 overrides the rank the right-hand side kernel sees: rank 0 adds xi_p on the linking rows, and a refinement step on a
// sharded handle passes every rank's PARTIAL residual of those rows (the all-reduce of the root right-hand side completes the sum)
static int enq_solve_local(tlpk_handle *h, const double *d_xip, const double *d_xid, int rhs_rank = -1) {
    {
        ProfScope ps(h, TLPK_KC_SPMV);
        // tickets + hand-over words of both sweeps back to all ones: the data is its own flag, ticket + 1 = 0 is the first item
        if (h->S.sweep && h->S.n_sweep_flags > 0) HIPCHK(h, hipMemsetAsync(h->d.sweep_tickets, 0xFF, (size_t)h->d.sweep_reset_bytes, h->stream));
        // (the give-up flag of an EARLIER solve must not fail this one; while the root front of an asynchronous update is still being factorised on its own stream the
        //  flag may be that update's: it stays)
        if (h->S.sweep && !h->root_pending) HIPCHK(h, hipMemsetAsync(h->d.ctx.info + 1, 0, sizeof(int), h->stream));
        if (h->S.system == 1) launch_k2_rhs(h->stream, h->d, h->S.k2_n, d_xip, d_xid, 0, rhs_rank >= 0 ? rhs_rank : h->opt.rank);        // [xi_d ; xi_p] permuted (sqd.jl:62-66)
        else launch_rhs(h->stream, h->d, h->d_D, d_xip, d_xid, rhs_rank >= 0 ? rhs_rank : h->opt.rank);
        launch_single_solve(h->stream, h->d);
    }
    run_launches(h, h->S.fwd_launches, 0, h->fwd_marker, 0);
    HIPCHK(h, hipGetLastError());
    return TLPK_OK;
}
static int enq_solve_finish(tlpk_handle *h, double *d_dx, double *d_dy, const double *d_xid) {
    if (h->root_pending) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_root, 0));      // the root front is being factorised on its own stream
    run_launches(h, h->S.fwd_launches, h->fwd_marker, h->S.fwd_launches.size(), 0);
    if (h->S.system == 1) { ProfScope ps(h, TLPK_KC_SPMV); launch_apply_signs(h->stream, h->d); }     // L S L' x = b: z = S y between the sweeps
    run_launches(h, h->S.bwd_launches, 0, h->S.bwd_launches.size(), 1);
    if (h->S.system == 1) {
        // multi-device mode: every shard stores the nodes it owns straight into the lead device's job-wide dx / dy
        ProfScope ps(h, TLPK_KC_SPMV);
        launch_k2_out(h->stream, h->d, h->S.k2_n, d_dx, h->shared_dy ? h->shared_dy : d_dy, 0, h->opt.rank, h->dx_local_only ? 1 : 0);
    } else {
        { ProfScope ps(h, TLPK_KC_SPMV); launch_unpermute(h->stream, h->d, d_dy, h->shared_dy, h->opt.rank); }
        { ProfScope ps(h, TLPK_KC_SPMV); launch_dx(h->stream, h->d, h->d_D, d_dy, d_xid, d_dx, h->dx_local_only ? 1 : 0); }
    }
    if (h->S.sweep) HIPCHK(h, hipMemcpyAsync(h->h_info + 1, h->d.ctx.info + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    return TLPK_OK;
}

int tlpk_solve_local(tlpk_handle *h, const double *d_xip, const double *d_xid) {
    if (!h || !d_xip || !d_xid) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (!h->factored) return TLPK_NOT_FACTORED;
    HIPCHK(h, hipSetDevice(h->device));
    prof_begin(h, false);
    h->solve_timed = false;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    h->solve_epoch += 1;
    if (h->solve_whole) return TLPK_OK;                  // tlpk_solve_device: the caller enqueues both halves (graph)
    if (int rc = enq_solve_local(h, d_xip, d_xid, h->rhs_all_ranks ? 0 : -1)) return rc;
    h->solve_local_done = true;
    return TLPK_OK;
}

int tlpk_root_rhs(tlpk_handle *h, double **d_ptr, int64_t *count) {
    if (!h || !d_ptr || !count) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    *d_ptr = h->nlink ? h->d.ctx.xw + h->first_link : nullptr;
    *count = h->nlink;
    return TLPK_OK;
}

int tlpk_solve_finish(tlpk_handle *h, double *d_dx, double *d_dy, const double *d_xid) {
    if (!h || !d_dx || !d_dy || !d_xid) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (!h->factored) return TLPK_NOT_FACTORED;
    if (!h->solve_local_done || h->refine_pending || h->pair_pending) { h->last_error = "tlpk_solve_finish without a preceding tlpk_solve_local"; return TLPK_BADARG; }
    h->solve_local_done = false;
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = enq_solve_finish(h, d_dx, d_dy, d_xid)) return rc;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipGetLastError());
    h->solve_timed = true;
    return TLPK_OK;
}

// one whole solve (both halves) from the graph cache, keyed by its four pointers
static int solve_whole(tlpk_handle *h, double *d_dx, double *d_dy, const double *d_xip, const double *d_xid) {
    if (!d_dx || !d_dy) return TLPK_BADARG;
    h->solve_whole = true;
    int rc = tlpk_solve_local(h, d_xip, d_xid);          // argument / state checks, ev0
    h->solve_whole = false;
    if (rc != TLPK_OK) return rc;
    const GraphKey key{2, {d_dx, d_dy, d_xip, d_xid}};
    rc = graph_or_direct(h, key, [&]() { const int q = enq_solve_local(h, d_xip, d_xid); return q != TLPK_OK ? q : enq_solve_finish(h, d_dx, d_dy, d_xid); }, graph_usable_solve(h));
    if (rc != TLPK_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipGetLastError());
    h->solve_timed = true;
    return TLPK_OK;
}

int tlpk_solve_device(tlpk_handle *h, double *d_dx, double *d_dy, const double *d_xip, const double *d_xid) {
    if (int g = sharded_needs_split(h, "tlpk_solve_device")) return g;
    const bool whole = h && h->sub.empty() && h->has_device && graph_usable_solve(h);
    int rc = whole ? solve_whole(h, d_dx, d_dy, d_xip, d_xid) : tlpk_solve_local(h, d_xip, d_xid);
    if (rc != TLPK_OK) return rc;
    if (!whole) rc = tlpk_solve_finish(h, d_dx, d_dy, d_xid);
    // optional iterative refinement on the residuals of the augmented system (KKT.jl:70-75): each step is one more solve with
    // (r1, r2) as right-hand side.  Off by default = the reference (spd.jl:68).  GUARDED (round 5): the candidate x + c is kept only if
    // |r1|inf shrinks and |r2|inf stays within 16 x of the unrefined solve's (kernels.hip: k_refine_decide) -- decided on the device, no host synchronisation; a rejected step ends the refinement of this solve
    // (tlpk_stats.refine_rejected).  On the north-star LP's late iterations an unguarded second step grew the dual residual from 2e-8 to 0.6
    // (profiles/r04_mpc_levers.txt): an option that exists must not make a solve worse.
    if (h->refine_steps > 0 && rc == TLPK_OK) {
        HIPCHK(h, hipMemsetAsync(h->d_ref, 0, 8 * sizeof(unsigned long long), h->stream));
        launch_residuals(h->stream, h->d, d_xip, d_xid, h->d_theta, h->d_regP, h->d_regD, d_dx, d_dy, h->d_r1, h->d_r2, 0);
        launch_absmax2(h->stream, h->d, h->d_r1, h->d_r2, h->d_ref + 0);      // |r1| current, |r2| of the unrefined solve
    }
    for (int it = 0; it < h->refine_steps && rc == TLPK_OK; ++it) {
        rc = whole ? solve_whole(h, h->d_cx, h->d_cy, h->d_r1, h->d_r2) : tlpk_solve_local(h, h->d_r1, h->d_r2);
        if (rc == TLPK_OK && !whole) rc = tlpk_solve_finish(h, h->d_cx, h->d_cy, h->d_r2);
        if (rc != TLPK_OK) break;
        launch_candidate(h->stream, h->S.n, d_dx, h->d_cx, h->S.m, d_dy, h->d_cy);
        // residuals of the candidate: the next step's right-hand side if the candidate is kept (after a rejection nothing is kept any more)
        launch_residuals(h->stream, h->d, d_xip, d_xid, h->d_theta, h->d_regP, h->d_regD, h->d_cx, h->d_cy, h->d_r1, h->d_r2, 0);
        launch_absmax2(h->stream, h->d, h->d_r1, h->d_r2, h->d_ref + 2);
        launch_refine_decide(h->stream, h->d_ref);
        launch_refine_commit(h->stream, h->S.n, d_dx, h->d_cx, h->S.m, d_dy, h->d_cy, h->d_ref);
        HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    }
    if (h->refine_steps > 0 && rc == TLPK_OK)
        HIPCHK(h, hipMemcpyAsync(h->h_info + 2, reinterpret_cast<int *>(h->d_ref + 4), sizeof(int), hipMemcpyDeviceToHost, h->stream));
    return rc;
}

// Iterative refinement in two halves, for sharded handles (and any single-rank K1 handle): one step is one more solve, so it holds
// one reduction of the root right-hand side, which the CALLER does between the halves --
//     tlpk_refine_local(h, dx, dy, xi_p, xi_d);  all-reduce(tlpk_root_rhs);  tlpk_refine_finish(h, dx, dy)
// after a completed solve (dx, dy in the layout tlpk_solve_finish leaves: a rank's own columns / block rows, the linking rows
// replicated).  Every rank forms the residuals it owns; on the linking rows its PARTIAL sum over its own columns (rank 0 adds
// xi_p - Rd dy), which the reduction inside the solve completes.  The buffers are allocated by the first call.
static int refine_buffers(tlpk_handle *h) {
    // (every buffer is checked, not only the first: an allocation that failed half way must not leave a later call with a null d_ref / d_cy -- round-5 advisor finding)
    const i64 nn = std::max<i64>(h->S.n, 1), mm = std::max<i64>(h->S.m, 1);
    int rc = TLPK_OK;
    if (!h->d_r1 && (rc = dev_alloc(h, &h->d_r1, mm)) != TLPK_OK) return rc;
    if (!h->d_r2 && (rc = dev_alloc(h, &h->d_r2, nn)) != TLPK_OK) return rc;
    if (!h->d_cx && (rc = dev_alloc(h, &h->d_cx, nn)) != TLPK_OK) return rc;
    if (!h->d_ref && (rc = dev_alloc(h, &h->d_ref, 8)) != TLPK_OK) return rc;
    if (!h->d_cy && (rc = dev_alloc(h, &h->d_cy, mm)) != TLPK_OK) return rc;
    return TLPK_OK;
}
int tlpk_refine_local(tlpk_handle *h, const double *d_dx, const double *d_dy, const double *d_xip, const double *d_xid) {
    if (!h || !d_dx || !d_dy || !d_xip || !d_xid) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (h->S.system == 1) { h->last_error = "iterative refinement: K1 only"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (!h->factored) return TLPK_NOT_FACTORED;
    if (h->solve_local_done) { h->last_error = "tlpk_refine_local inside an unfinished solve"; return TLPK_BADARG; }
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = refine_buffers(h)) return rc;
    prof_begin(h, false);
    h->solve_timed = false;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    h->solve_epoch += 1;
    // (rhs_all_ranks: the device-resident loops' convention -- this shard's xi_p is its PARTIAL of the linking rows and counts whatever the rank)
    launch_residuals(h->stream, h->d, d_xip, d_xid, h->d_theta, h->d_regP, h->d_regD, d_dx, d_dy, h->d_r1, h->d_r2, h->opt.rank, h->rhs_all_ranks ? 1 : 0);
    if (int rc = enq_solve_local(h, h->d_r1, h->d_r2, 0)) return rc;
    h->solve_local_done = true; h->refine_pending = true;
    return TLPK_OK;
}
int tlpk_refine_finish(tlpk_handle *h, double *d_dx, double *d_dy) {
    if (!h || !d_dx || !d_dy) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (!h->solve_local_done || !h->refine_pending) { h->last_error = "tlpk_refine_finish without a preceding tlpk_refine_local"; return TLPK_BADARG; }
    h->solve_local_done = false; h->refine_pending = false;
    HIPCHK(h, hipSetDevice(h->device));
    double *keep = h->shared_dy; const bool keep_lo = h->dx_local_only;
    h->shared_dy = nullptr; h->dx_local_only = false;                     // the correction stays rank-local
    const int rc = enq_solve_finish(h, h->d_cx, h->d_cy, h->d_r2);
    h->shared_dy = keep; h->dx_local_only = keep_lo;
    if (rc != TLPK_OK) return rc;
    launch_axpy2(h->stream, h->S.n, d_dx, h->d_cx, h->S.m, d_dy, h->d_cy);
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipGetLastError());
    h->solve_timed = true;
    return TLPK_OK;
}

// the two halves of a PAIR of solves (two right-hand sides in one pass over L), split like enq_solve_local / enq_solve_finish at the
// reduction of the root right-hand sides (two of them: tlpk_root_rhs and tlpk_root_rhs2)
static int enq_solve2_local(tlpk_handle *h, const double *const *xip, const double *const *xid, int rhs_rank) {
    const bool k2 = h->S.system == 1;
    const int rank = rhs_rank >= 0 ? rhs_rank : h->opt.rank;
    {
        ProfScope ps(h, TLPK_KC_SPMV);
        if (h->S.n_sweep_flags > 0) HIPCHK(h, hipMemsetAsync(h->d.sweep_tickets, 0xFF, (size_t)h->d.sweep_reset_bytes2, h->stream));
        if (k2) {
            for (int r = 0; r < 2; ++r) launch_k2_rhs(h->stream, h->d, h->S.k2_n, xip[r], xid[r], r, rank);
        } else launch_rhs2(h->stream, h->d, h->d_D, xip, xid, rank);          // both right-hand sides in one launch each (round 6)
        launch_single_solve(h->stream, h->d, 2);
    }
    run_launches(h, h->S.fwd_launches, 0, h->fwd_marker, 0, 2);
    HIPCHK(h, hipGetLastError());
    return TLPK_OK;
}
static int enq_solve2_finish(tlpk_handle *h, double *const *dx, double *const *dy, const double *const *xid) {
    const bool k2 = h->S.system == 1;
    if (h->root_pending) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_root, 0));      // the root front is being factorised on its own stream
    run_launches(h, h->S.fwd_launches, h->fwd_marker, h->S.fwd_launches.size(), 0, 2);
    if (k2) { ProfScope ps(h, TLPK_KC_SPMV); launch_apply_signs(h->stream, h->d, 0); launch_apply_signs(h->stream, h->d, 1); }
    run_launches(h, h->S.bwd_launches, 0, h->S.bwd_launches.size(), 1, 2);
    {
        ProfScope ps(h, TLPK_KC_SPMV);
        if (k2) { for (int r = 0; r < 2; ++r) launch_k2_out(h->stream, h->d, h->S.k2_n, dx[r], dy[r], r, h->opt.rank, 0); }
        else if (h->shared_dy || h->dx_local_only) {                              // (shards of a multi-device handle publish into the lead's vectors: per right-hand side)
            for (int r = 0; r < 2; ++r) {
                launch_unpermute(h->stream, h->d, dy[r], nullptr, h->opt.rank, r);
                launch_dx(h->stream, h->d, h->d_D, dy[r], xid[r], dx[r], 0);
            }
        } else {
            launch_unpermute2(h->stream, h->d, dy, h->opt.rank);
            launch_dx2(h->stream, h->d, h->d_D, dy, xid, dx);
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->h_info + 1, h->d.ctx.info + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    return TLPK_OK;
}

// Two right-hand sides against the same factor in ONE pass over L (the persistent sweeps are bound by the bytes of L: the
// pair costs little more than one solve).  Tulip's HSD step has such a pair in every iteration: the h-system and the predictor
// (/root/reference/src/IPM/HSD/step.jl:63 and :79 -- neither right-hand side depends on the other solve).  Results are bit-identical
// to two tlpk_solve_device calls.  Single-rank handles; with refine_steps > 0 the pair falls back to two refined solves.
int tlpk_solve2_device(tlpk_handle *h, double *d_dx0, double *d_dy0, const double *d_xip0, const double *d_xid0,
                       double *d_dx1, double *d_dy1, const double *d_xip1, const double *d_xid1) {
    if (!h || !d_dx0 || !d_dy0 || !d_xip0 || !d_xid0 || !d_dx1 || !d_dy1 || !d_xip1 || !d_xid1) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (int g = sharded_needs_split(h, "tlpk_solve2_device")) return g;
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (!h->factored) return TLPK_NOT_FACTORED;
    if (h->refine_steps > 0 || !h->S.sweep) {            // refinement / launch-per-block schedule: two ordinary solves
        const int rc = tlpk_solve_device(h, d_dx0, d_dy0, d_xip0, d_xid0);
        return rc != TLPK_OK ? rc : tlpk_solve_device(h, d_dx1, d_dy1, d_xip1, d_xid1);
    }
    HIPCHK(h, hipSetDevice(h->device));
    prof_begin(h, false);
    h->solve_timed = false;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    const double *xip[2] = {d_xip0, d_xip1}, *xid[2] = {d_xid0, d_xid1};
    double *dx[2] = {d_dx0, d_dx1}, *dy[2] = {d_dy0, d_dy1};
    h->solve_epoch += 1;
    const GraphKey key{3, {d_dx0, d_dy0, d_xip0, d_xid0, d_dx1, d_dy1, d_xip1, d_xid1}};
    const int grc = graph_or_direct(h, key, [&]() -> int {
        const int q = enq_solve2_local(h, xip, xid, -1);
        return q != TLPK_OK ? q : enq_solve2_finish(h, dx, dy, xid);
    }, graph_usable_solve(h));
    if (grc != TLPK_OK) return grc;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipGetLastError());
    h->solve_timed = true;
    return TLPK_OK;
}

// The pair in two halves, for sharded handles: tlpk_solve2_local -> all-reduce of tlpk_root_rhs AND tlpk_root_rhs2 (the root right-hand
// sides of the two systems; one collective over both buffers if the communicator allows) -> tlpk_solve2_finish.
int tlpk_solve2_local(tlpk_handle *h, const double *d_xip0, const double *d_xid0, const double *d_xip1, const double *d_xid1) {
    if (!h || !d_xip0 || !d_xid0 || !d_xip1 || !d_xid1) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (!h->factored) return TLPK_NOT_FACTORED;
    if (!h->S.sweep) { h->last_error = "tlpk_solve2_local needs the persistent-sweep schedule (TLPK_SWEEP=0 is set)"; return TLPK_BADARG; }
    if (h->solve_local_done) { h->last_error = "tlpk_solve2_local inside an unfinished solve"; return TLPK_BADARG; }
    HIPCHK(h, hipSetDevice(h->device));
    prof_begin(h, false);
    h->solve_timed = false;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    h->solve_epoch += 1;
    const double *xip[2] = {d_xip0, d_xip1}, *xid[2] = {d_xid0, d_xid1};
    if (int rc = enq_solve2_local(h, xip, xid, h->rhs_all_ranks ? 0 : -1)) return rc;
    h->solve_local_done = true; h->pair_pending = true;
    return TLPK_OK;
}
int tlpk_root_rhs2(tlpk_handle *h, double **d_ptr, int64_t *count) {
    if (!h || !d_ptr || !count) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    *d_ptr = h->nlink ? h->d.ctx.xw + h->d.ctx.xw2 + h->first_link : nullptr;
    *count = h->nlink;
    return TLPK_OK;
}
int tlpk_solve2_finish(tlpk_handle *h, double *d_dx0, double *d_dy0, const double *d_xid0, double *d_dx1, double *d_dy1, const double *d_xid1) {
    if (!h || !d_dx0 || !d_dy0 || !d_xid0 || !d_dx1 || !d_dy1 || !d_xid1) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (!h->solve_local_done || !h->pair_pending) { h->last_error = "tlpk_solve2_finish without a preceding tlpk_solve2_local"; return TLPK_BADARG; }
    h->solve_local_done = false; h->pair_pending = false;
    HIPCHK(h, hipSetDevice(h->device));
    double *dx[2] = {d_dx0, d_dx1}, *dy[2] = {d_dy0, d_dy1};
    const double *xid[2] = {d_xid0, d_xid1};
    if (int rc = enq_solve2_finish(h, dx, dy, xid)) return rc;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipGetLastError());
    h->solve_timed = true;
    return TLPK_OK;
}

int tlpk_sync(tlpk_handle *h) {
    if (!h) return TLPK_BADARG;
    if (!h->sub.empty()) { int w = TLPK_OK; for (tlpk_handle *c : h->sub) { const int rc = tlpk_sync(c); if (rc != TLPK_OK) w = rc; } return w; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    HIPCHK(h, hipSetDevice(h->device));
    int pending_rc = TLPK_OK;
    if (h->root_pending) { pending_rc = update_async_wait(h); h->solve_timed = false; }      // verdict of tlpk_update_device_async
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (pending_rc != TLPK_OK) { (void)hipGetLastError(); prof_collect(h); return pending_rc; }
    if (h->solve_timed) {               // ev0/ev1 bracket a complete solve only then
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->ms_solve = ms;
        h->solve_timed = false;
    }
    (void)hipGetLastError();
    prof_collect(h);
    if (h->refine_steps > 0) h->refine_rejected = h->h_info[2];
    if (h->h_info[1] != 0) {
        h->last_error = "a solve sweep gave up waiting for a block hand-over (internal scheduling error); results are invalid";
        return TLPK_INTERNAL;
    }
    return TLPK_OK;
}

void *tlpk_stream(tlpk_handle *h) { return h ? (void *)(h->sub.empty() ? h->stream : h->sub[0]->stream) : nullptr; }

int tlpk_solve(tlpk_handle *h, double *dx, double *dy, const double *xi_p, const double *xi_d) {
    if (!h || !dx || !dy || !xi_p || !xi_d) return TLPK_BADARG;
    if (!h->sub.empty()) return multi_solve(h, dx, dy, xi_p, xi_d);
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (!h->factored) return TLPK_NOT_FACTORED;
    if (int g = sharded_needs_split(h, "tlpk_solve")) return g;
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = ensure_pinned(h)) return rc;
    const i64 un = user_n(h), um = user_m(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    static const bool timing = [] { const char *e = std::getenv("TLPK_HOSTIO_TIMING"); return e && std::atoi(e) != 0; }();
    h->io_timing = timing;
    const double t0 = timing ? now_ms() : 0.0;
    const HostVec in[2] = {{h->d_xip, const_cast<double *>(xi_p), um}, {h->d_xid, const_cast<double *>(xi_d), un}};
    if (int rc = stage_in(h, in, 2)) return rc;
    const double t1 = timing ? now_ms() : 0.0;
    int rc = tlpk_solve_device(h, h->d_dx, h->d_dy, h->d_xip, h->d_xid);
    if (rc != TLPK_OK) return rc;
    const double t2 = timing ? now_ms() : 0.0;
    // dy is final before k_dx starts: its pieces go first
    const HostVec out[2] = {{h->d_dy, dy, um}, {h->d_dx, dx, un}};
    rc = stage_out(h, out, 2);
    const int src = tlpk_sync(h);                          // status words (a sweep that gave up waiting), timers; the stream is idle by now
    if (timing) std::fprintf(stderr, "tlpk_solve host path: stage in + H2D issued %.3f ms | kernels enqueued %.3f ms | first D2H group landed +%.3f ms | last D2H group landed +%.3f ms | "
                             "copied out +%.3f ms | device solve (events) %.3f ms | total %.3f ms\n",
                             t1 - t0, t2 - t1, h->io_t_first - t2, h->io_t_last - h->io_t_first, now_ms() - h->io_t_last, h->ms_solve, now_ms() - t0);
    return rc != TLPK_OK ? rc : src;
}

// ---- single-process multi-device mode (block-angular LPs; SURVEY.md 8e "one-process-8-devices") ----
// One handle drives `ngpus` devices from one host thread (what a Julia process needs): internally one sharded
// handle per device (rank r of ngpus, same split-phase schedule as the one-process-per-GPU mode).  The two
// reductions of a Newton step -- the root (linking) panel after the local factorisations, the root right-hand
// side inside every solve -- are done by the library, stream-ordered through events, no host synchronisation
// between the halves:
//   * default: peer-to-peer copies -- gather on the lead device, add in rank order (deterministic, bit-identical to the
//     one-process-per-GPU mode with an ordered reduction), copy back;
//   * TLPK_MULTI_REDUCE=rccl: ncclAllReduce over xGMI on every shard's stream (librccl.so is dlopen'ed, one communicator per
//     device from ncclCommInitAll); needs distinct devices; the sum order is RCCL's.
// Results are written by every rank straight into the lead device's dx / dy (P2P stores of the entries it owns) and leave
// through one pinned copy.  Host -> device: every shard receives only the slices of theta / regP / regD / xi it reads.
namespace {

// --- RCCL through dlopen: the library does not link against it (single-GPU users never load it) ---
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load(std::string &err) {
        if (lib) return true;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { err = std::string("dlopen(librccl.so): ") + dlerror(); return false; }
        *(void **)&CommInitAll = dlsym(lib, "ncclCommInitAll"); *(void **)&CommDestroy = dlsym(lib, "ncclCommDestroy");
        *(void **)&AllReduce = dlsym(lib, "ncclAllReduce"); *(void **)&GroupStart = dlsym(lib, "ncclGroupStart");
        *(void **)&GroupEnd = dlsym(lib, "ncclGroupEnd"); *(void **)&GetErrorString = dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd) { err = "librccl.so lacks the expected symbols"; return false; }
        return true;
    }
};
Rccl g_rccl;
constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0;          // rccl.h: ncclFloat64 = 8, ncclSum = 0

// which: 0 root panel, 1 root right-hand side, 2 root right-hand side of the second system of a pair
int root_buf(tlpk_handle *c, int which, double **p, int64_t *cnt) {
    return which == 0 ? tlpk_root_panel(c, p, cnt) : (which == 1 ? tlpk_root_rhs(c, p, cnt) : tlpk_root_rhs2(c, p, cnt));
}
int multi_allreduce_rccl(tlpk_handle *h, int which) {
    const int N = (int)h->sub.size();
    int rc = g_rccl.GroupStart();
    for (int r = 0; r < N && rc == 0; ++r) {
        tlpk_handle *c = h->sub[r];
        double *p = nullptr; int64_t cnt = 0;
        const int q = root_buf(c, which, &p, &cnt);
        if (q != TLPK_OK) { g_rccl.GroupEnd(); return q; }
        if (cnt == 0) { g_rccl.GroupEnd(); return TLPK_OK; }
        HIPCHK(h, hipSetDevice(c->device));
        rc = g_rccl.AllReduce(p, p, (size_t)cnt, NCCL_DOUBLE, NCCL_SUM, h->multi_comm[r], c->stream);      // in place, on the shard's stream
    }
    const int rc2 = g_rccl.GroupEnd();
    if (rc != 0 || rc2 != 0) {
        h->last_error = std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc ? rc : rc2) : "error");
        return TLPK_HIPERR;
    }
    return TLPK_OK;
}

// Reduce-scatter + all-gather over peer copies (round 4; the default): shard r owns slice r of the buffer.  Every shard sends the OTHER shards'
// slices of its buffer to their owners (its own stream, after its local work), the owner adds the N contributions to its slice in RANK order
// (k_sum_ranked: bitwise the same sum whoever owns the slice) and sends the reduced slice to every peer.  2 (N - 1) copies of 1/N of the
// buffer per shard, all of them between different pairs of devices at the same time -- the gather-to-lead form moved 2 (N - 1) whole buffers
// through the lead's links one after the other.  Deterministic; exercisable with several shards on one GPU.
int multi_allreduce_rs(tlpk_handle *h, int which) {
    const int N = (int)h->sub.size();
    double *p[MAX_DEVICES]; int64_t cnt = 0;
    for (int r = 0; r < N; ++r) {
        int64_t cr = 0;
        const int rc = root_buf(h->sub[(size_t)r], which, &p[r], &cr);
        if (rc != TLPK_OK) return rc;
        if (r == 0) cnt = cr; else if (cr != cnt) { h->last_error = "root buffers of the ranks differ"; return TLPK_INTERNAL; }
    }
    if (cnt == 0) return TLPK_OK;
    const i64 sl = ((cnt + N - 1) / N + 15) / 16 * 16;                                   // slice length: multiples of 16 doubles
    if (sl > h->rs_slice) { h->last_error = "reduce-scatter: buffer larger than the staging slices sized at create"; return TLPK_INTERNAL; }
    auto lo = [&](int r) { return std::min<i64>(cnt, (i64)r * sl); };
    auto len = [&](int r) { return std::min<i64>(cnt, (i64)(r + 1) * sl) - lo(r); };
    for (int s_ = 0; s_ < N; ++s_) {                    // scatter: shard s sends slice r of its buffer to shard r
        tlpk_handle *c = h->sub[(size_t)s_];
        HIPCHK(h, hipSetDevice(c->device));
        for (int r = 0; r < N; ++r) {
            if (r == s_ || len(r) <= 0) continue;
            tlpk_handle *o = h->sub[(size_t)r];
            HIPCHK(h, hipMemcpyPeerAsync(o->rs_stage + (size_t)(s_ < r ? s_ : s_ - 1) * (size_t)h->rs_slice, o->device, p[s_] + lo(r), c->device, (size_t)len(r) * 8, c->stream));
        }
        HIPCHK(h, hipEventRecord(h->multi_ev[s_], c->stream));
    }
    for (int r = 0; r < N; ++r) {                       // reduce the own slice, send it to everybody
        tlpk_handle *c = h->sub[(size_t)r];
        HIPCHK(h, hipSetDevice(c->device));
        for (int s_ = 0; s_ < N; ++s_) if (s_ != r) HIPCHK(h, hipStreamWaitEvent(c->stream, h->multi_ev[s_], 0));
        if (len(r) > 0) {
            launch_sum_ranked(c->stream, len(r), p[r] + lo(r), c->rs_stage, N, r, h->rs_slice);
            for (int t = 0; t < N; ++t)
                if (t != r) HIPCHK(h, hipMemcpyPeerAsync(p[t] + lo(r), h->sub[(size_t)t]->device, p[r] + lo(r), c->device, (size_t)len(r) * 8, c->stream));
        }
        HIPCHK(h, hipEventRecord(h->multi_ev2[r], c->stream));
    }
    for (int t = 0; t < N; ++t) {                       // nobody goes on before every slice of its buffer has arrived
        tlpk_handle *c = h->sub[(size_t)t];
        HIPCHK(h, hipSetDevice(c->device));
        for (int r = 0; r < N; ++r) if (r != t) HIPCHK(h, hipStreamWaitEvent(c->stream, h->multi_ev2[r], 0));
    }
    HIPCHK(h, hipSetDevice(h->sub[0]->device));
    return TLPK_OK;
}

int multi_allreduce(tlpk_handle *h, int which) {
    if (h->multi_rccl) return multi_allreduce_rccl(h, which);
    if (h->multi_mode == 1 && h->sub.size() > 1) return multi_allreduce_rs(h, which);
    tlpk_handle *lead = h->sub[0];
    double *p0 = nullptr; int64_t cnt = 0;
    int rc = root_buf(lead, which, &p0, &cnt);
    if (rc != TLPK_OK || cnt == 0) return rc;
    const int N = (int)h->sub.size();
    for (int r = 1; r < N; ++r) {                        // gather: every peer sends on its own stream, after its local work
        tlpk_handle *c = h->sub[r];
        double *pr = nullptr; int64_t cr = 0;
        rc = root_buf(c, which, &pr, &cr);
        if (rc != TLPK_OK || cr != cnt) { h->last_error = "root buffers of the ranks differ"; return TLPK_INTERNAL; }
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipMemcpyPeerAsync(h->multi_tmp + (size_t)(r - 1) * (size_t)cnt, lead->device, pr, c->device, (size_t)cnt * 8, c->stream));
        HIPCHK(h, hipEventRecord(h->multi_ev[r], c->stream));
    }
    HIPCHK(h, hipSetDevice(lead->device));
    for (int r = 1; r < N; ++r) HIPCHK(h, hipStreamWaitEvent(lead->stream, h->multi_ev[r], 0));
    // The sum goes to its own buffer: the lead continues with ITS copy (it factorises / solves the root in place)
    // while the peers still read the result.  The buffer is reused by the next reduction, which every peer enters
    // only after its copy below (stream order) and the lead only after all peers' gathers (events).
    double *red = h->multi_tmp + h->multi_red_off;
    launch_sum_to(lead->stream, cnt, red, p0, h->multi_tmp, N - 1, cnt);        // rank order: deterministic
    HIPCHK(h, hipEventRecord(h->multi_done, lead->stream));
    HIPCHK(h, hipMemcpyAsync(p0, red, (size_t)cnt * 8, hipMemcpyDeviceToDevice, lead->stream));
    for (int r = 1; r < N; ++r) {                        // copy back, on the peer's stream
        tlpk_handle *c = h->sub[r];
        double *pr = nullptr; int64_t cr = 0;
        rc = root_buf(c, which, &pr, &cr);
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipStreamWaitEvent(c->stream, h->multi_done, 0));
        HIPCHK(h, hipMemcpyPeerAsync(pr, c->device, red, lead->device, (size_t)cnt * 8, c->stream));
    }
    return TLPK_OK;
}


// Guarded refinement on a multi-device handle (round 5).  |r1|inf and |r2|inf of the augmented system's residuals for shard-resident
// solutions (the rule of kernels.hip: k_refine_decide, evaluated on the host): every shard forms the residuals of the rows / columns it owns and its PARTIAL sums on the linking rows (same convention as the
// right-hand side of the solve: `all_ranks` = every shard's xi_p counts there, otherwise rank 0's only); the owned maxima are reduced on the
// devices, the linking rows are summed on the host in shard order.  Host-synchronised: refinement is an off-by-default option.
int refine_buffers_multi(tlpk_handle *c) {
    if (int rc = refine_buffers(c)) return rc;
    if (c->d_bx) return TLPK_OK;
    if (int rc = dev_alloc(c, &c->d_bx, std::max<i64>(c->S.n, 1))) return rc;
    return dev_alloc(c, &c->d_by, std::max<i64>(c->S.m, 1));
}
int multi_resid_norm(tlpk_handle *h, double *const *dx, double *const *dy, const double *const *xip, const double *const *xid, bool all_ranks, double *out /* |r1|, |r2| */) {
    const int N = (int)h->sub.size();
    for (int r = 0; r < N; ++r) {
        tlpk_handle *c = h->sub[(size_t)r];
        HIPCHK(h, hipSetDevice(c->device));
        if (int rc = refine_buffers_multi(c)) { h->last_error = c->last_error; return rc; }
        HIPCHK(h, hipMemsetAsync(c->d_ref, 0, 8 * sizeof(unsigned long long), c->stream));
        launch_residuals(c->stream, c->d, xip[r], xid[r], c->d_theta, c->d_regP, c->d_regD, dx[r], dy[r], c->d_r1, c->d_r2, c->opt.rank, all_ranks ? 1 : 0);
        launch_absmax2(c->stream, c->d, c->d_r1, c->d_r2, c->d_ref, 1);
    }
    double n1 = 0.0, n2 = 0.0;
    std::vector<double> link, part;
    for (int r = 0; r < N; ++r) {
        tlpk_handle *c = h->sub[(size_t)r];
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipStreamSynchronize(c->stream));
        unsigned long long bits[2] = {0, 0};
        HIPCHK(h, hipMemcpy(bits, c->d_ref, sizeof(bits), hipMemcpyDeviceToHost));
        double v1, v2; std::memcpy(&v1, &bits[0], 8); std::memcpy(&v2, &bits[1], 8);
        if (!(v1 <= n1)) n1 = v1;                                  // (a NaN pattern propagates)
        if (!(v2 <= n2)) n2 = v2;
        const i64 span = c->link_hi - c->link_lo;
        if (span > 0) {
            part.resize((size_t)span);
            HIPCHK(h, hipMemcpy(part.data(), c->d_r1 + c->link_lo, (size_t)span * 8, hipMemcpyDeviceToHost));
            if (link.empty()) link.assign((size_t)span, 0.0);
            for (i64 i = 0; i < span; ++i) if (c->S.row_local[(size_t)(c->link_lo + i)] == 2) link[(size_t)i] += part[(size_t)i];
        }
    }
    for (double v : link) { const double a = std::fabs(v); if (!(a <= n1)) n1 = a; }
    HIPCHK(h, hipSetDevice(h->sub[0]->device));
    out[0] = n1; out[1] = n2;
    return TLPK_OK;
}
// one guarded step: backup, step(), verdict; *stop = the step was rejected (the iterate is restored)
int multi_guarded_step(tlpk_handle *h, double *const *dx, double *const *dy, const double *const *xip, const double *const *xid, bool all_ranks,
                       double *norm, bool *stop, const std::function<int()> &step) {
    for (size_t r = 0; r < h->sub.size(); ++r) {
        tlpk_handle *c = h->sub[r];
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipMemcpyAsync(c->d_bx, dx[r], (size_t)c->S.n * 8, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(h, hipMemcpyAsync(c->d_by, dy[r], (size_t)c->S.m * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    if (int rc = step()) return rc;
    double after[2] = {0.0, 0.0};
    if (int rc = multi_resid_norm(h, dx, dy, xip, xid, all_ranks, after)) return rc;
    if (after[0] < norm[0] && after[1] <= 16.0 * norm[1]) { norm[0] = after[0]; *stop = false; return TLPK_OK; }      // the rule of k_refine_decide
    for (size_t r = 0; r < h->sub.size(); ++r) {
        tlpk_handle *c = h->sub[r];
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipMemcpyAsync(dx[r], c->d_bx, (size_t)c->S.n * 8, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(h, hipMemcpyAsync(dy[r], c->d_by, (size_t)c->S.m * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    h->refine_rejected += 1; *stop = true;
    return TLPK_OK;
}

// host -> device copy of the entries [lo, hi) of a pinned full-length vector (device arrays are full length too)
inline hipError_t upload_range(double *dst, const double *src, i64 lo, i64 hi, hipStream_t st) {
    return hi > lo ? hipMemcpyAsync(dst + lo, src + lo, (size_t)(hi - lo) * 8, hipMemcpyHostToDevice, st) : hipSuccess;
}

void multi_comm_destroy(void *comm) { if (g_rccl.CommDestroy) g_rccl.CommDestroy(comm); }

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// second half of an update: reduction of the root panel, every shard's root front, the verdict
int multi_update_tail(tlpk_handle *h, double t_in) {
    if (int rc = multi_allreduce(h, 0)) return rc;
    if (int rc = for_shards(h, [](tlpk_handle *c, int) { return update_finish_enqueue(c); })) return rc;      // every root front is enqueued before anybody waits
    h->ms_enqueue_update = now_ms() - t_in;
    int worst = TLPK_OK; h->fail_col = -1;
    for (tlpk_handle *c : h->sub) {
        const int rc = update_finish_wait(c);
        if (rc == TLPK_NOT_POSDEF) { if (h->fail_col < 0 || c->fail_col < h->fail_col) h->fail_col = c->fail_col; if (worst == TLPK_OK) worst = rc; }
        else if (rc != TLPK_OK) { h->last_error = c->last_error; worst = rc; }
    }
    h->ms_update = now_ms() - t_in;
    h->factored = (worst == TLPK_OK);
    return worst;
}

int multi_update(tlpk_handle *h, const double *theta, const double *regP, const double *regD) {
    tlpk_handle *lead = h->sub[0];
    const i64 n = h->S.n, m = h->S.m;
    const double t_in = now_ms();
    HIPCHK(h, hipSetDevice(lead->device));
    if (int rc = ensure_pinned(lead)) { h->last_error = lead->last_error; return rc; }
    for (tlpk_handle *c : h->sub) { HIPCHK(h, hipSetDevice(c->device)); HIPCHK(h, hipStreamSynchronize(c->stream)); }   // staging area free again
    double *p0 = lead->pin_in, *p1 = p0 + n, *p2 = p1 + n;
    host_copy(p0, theta, n, true); host_copy(p1, regP, n, true); host_copy(p2, regD, m, true);
    if (int rc = for_shards(h, [&](tlpk_handle *c, int) -> int {
            // only what this shard reads: its columns of theta / regP, its block rows and the linking rows of regD
            HIPCHK(c, upload_range(c->d_theta, p0, c->col_lo, c->col_hi, c->stream));
            HIPCHK(c, upload_range(c->d_regP, p1, c->col_lo, c->col_hi, c->stream));
            HIPCHK(c, upload_range(c->d_regD, p2, c->row_lo, c->row_hi, c->stream));
            HIPCHK(c, upload_range(c->d_regD, p2, c->link_lo, c->link_hi, c->stream));
            return tlpk_update_local(c, c->d_theta, c->d_regP, c->d_regD);
        })) return rc;
    return multi_update_tail(h, t_in);
}

int multi_solve(tlpk_handle *h, double *dx, double *dy, const double *xi_p, const double *xi_d) {
    if (!h->factored) return TLPK_NOT_FACTORED;
    tlpk_handle *lead = h->sub[0];
    const i64 n = h->S.n, m = h->S.m;
    HIPCHK(h, hipSetDevice(lead->device));
    if (int rc = ensure_pinned(lead)) { h->last_error = lead->last_error; return rc; }
    for (tlpk_handle *c : h->sub) { HIPCHK(h, hipSetDevice(c->device)); HIPCHK(h, hipStreamSynchronize(c->stream)); }
    double *pi0 = lead->pin_in, *pi1 = pi0 + m;
    host_copy(pi0, xi_p, m, true); host_copy(pi1, xi_d, n, true);
    if (int rc = for_shards(h, [&](tlpk_handle *c, int) -> int {
            HIPCHK(c, upload_range(c->d_xip, pi0, c->row_lo, c->row_hi, c->stream));
            HIPCHK(c, upload_range(c->d_xip, pi0, c->link_lo, c->link_hi, c->stream));
            HIPCHK(c, upload_range(c->d_xid, pi1, c->col_lo, c->col_hi, c->stream));
            return tlpk_solve_local(c, c->d_xip, c->d_xid);
        })) return rc;
    if (int rc = multi_allreduce(h, 1)) return rc;
    if (h->refine_steps > 0) {
        // iterative refinement: every shard keeps its solution rank-local (own columns / block rows, linking rows replicated), each
        // step is one more split solve on the residuals with the same reduction in the middle, and the owned slices are published
        // to the lead device's job-wide vectors at the end
        for (tlpk_handle *c : h->sub) {
            c->shared_dy = nullptr; c->dx_local_only = false;
            const int rc = tlpk_solve_finish(c, c->d_dx, c->d_dy, c->d_xid);
            if (rc != TLPK_OK) { h->last_error = c->last_error; return rc; }
        }
        {
            const int N = (int)h->sub.size();
            double *dxs[MAX_DEVICES], *dys[MAX_DEVICES]; const double *xps[MAX_DEVICES], *xds[MAX_DEVICES];
            for (int r = 0; r < N; ++r) { tlpk_handle *c = h->sub[(size_t)r]; dxs[r] = c->d_dx; dys[r] = c->d_dy; xps[r] = c->d_xip; xds[r] = c->d_xid; }
            double norm[2] = {0.0, 0.0}; bool stop = false;      // |r1| of the current iterate, |r2| of the unrefined solve
            h->refine_rejected = 0;
            if (int rc = multi_resid_norm(h, dxs, dys, xps, xds, false, norm)) return rc;
            for (int it = 0; it < h->refine_steps && !stop; ++it) {
                if (int rc = multi_guarded_step(h, dxs, dys, xps, xds, false, norm, &stop, [&]() -> int {
                        for (tlpk_handle *c : h->sub) {
                            const int q = tlpk_refine_local(c, c->d_dx, c->d_dy, c->d_xip, c->d_xid);
                            if (q != TLPK_OK) { h->last_error = c->last_error; return q; }
                        }
                        if (int q = multi_allreduce(h, 1)) return q;
                        for (tlpk_handle *c : h->sub) {
                            const int q = tlpk_refine_finish(c, c->d_dx, c->d_dy);
                            if (q != TLPK_OK) { h->last_error = c->last_error; return q; }
                        }
                        return TLPK_OK;
                    })) return rc;
            }
        }
        // the lead's own last kernels still read-modify-write ITS rank-local dx / dy (= the job-wide vectors): peers publish after them
        HIPCHK(h, hipSetDevice(lead->device));
        HIPCHK(h, hipEventRecord(h->multi_done, lead->stream));
        for (size_t r = 1; r < h->sub.size(); ++r) {
            tlpk_handle *c = h->sub[r];
            HIPCHK(h, hipSetDevice(c->device));
            HIPCHK(h, hipStreamWaitEvent(c->stream, h->multi_done, 0));
            launch_publish(c->stream, c->d, c->d_dx, lead->d_dx, c->d_dy, lead->d_dy);      // owned columns / block rows only (P2P stores)
            HIPCHK(h, hipEventRecord(h->multi_ev[r], c->stream));
        }
    } else if (int rc = for_shards(h, [&](tlpk_handle *c, int r) -> int {
        // every rank fills its own entries of the lead device's dx / dy (P2P stores); its local dy feeds its k_dx
        c->shared_dy = lead->d_dy; c->dx_local_only = true;
        const int q = tlpk_solve_finish(c, lead->d_dx, (c->S.system == 1) ? lead->d_dy : (r == 0 ? h->multi_tmp + h->multi_dy0_off : c->d_dy), c->d_xid);
        if (q != TLPK_OK) return q;
        if (r > 0) HIPCHK(c, hipEventRecord(h->multi_ev[r], c->stream));
        return TLPK_OK;
    })) return rc;
    HIPCHK(h, hipSetDevice(lead->device));
    for (size_t r = 1; r < h->sub.size(); ++r) HIPCHK(h, hipStreamWaitEvent(lead->stream, h->multi_ev[r], 0));
    double *po0 = lead->pin_out, *po1 = po0 + m;
    HIPCHK(h, hipMemcpyAsync(po0, lead->d_dy, (size_t)m * 8, hipMemcpyDeviceToHost, lead->stream));
    HIPCHK(h, hipMemcpyAsync(po1, lead->d_dx, (size_t)n * 8, hipMemcpyDeviceToHost, lead->stream));
    int worst = TLPK_OK;
    for (tlpk_handle *c : h->sub) { const int rc = tlpk_sync(c); if (rc != TLPK_OK) { h->last_error = c->last_error; worst = rc; } }
    if (worst != TLPK_OK) return worst;
    host_copy(dy, po0, m, false); host_copy(dx, po1, n, false);
    return TLPK_OK;
}

// index ranges a shard reads from the job-wide input vectors
void shard_ranges(tlpk_handle *c) {
    const Symbolic &S = c->S;
    if (S.system == 1) {
        // K2: the Symbolic describes the augmented matrix -- node j < n is user column j, node n + i user row i; row_local of a node = 0 (another
        // shard's), 1 (this shard's block) or 2 (replicated root front).  A shard reads theta / regP / xi_d of the variable nodes it owns (the lead
        // also those of the root front: everything assembled into the root comes from rank 0), regD / xi_p of its constraint nodes and of the
        // linking constraints (round 3 uploaded the whole user vectors to every shard)
        const i64 n = S.k2_n, mm = S.k2_m;
        auto span = [&](i64 a, i64 b, auto pred, i64 &lo, i64 &hi) {
            lo = hi = 0; bool any = false;
            for (i64 v = a; v < b; ++v) if (pred(S.row_local[(size_t)v])) { if (!any) { lo = v - a; any = true; } hi = v - a + 1; }
        };
        const bool lead = S.front_local.empty() ? true : (c->opt.rank == 0);
        span(0, n, [&](char f) { return f == 1 || (f == 2 && lead); }, c->col_lo, c->col_hi);
        span(n, n + mm, [](char f) { return f == 1; }, c->row_lo, c->row_hi);
        span(n, n + mm, [](char f) { return f == 2; }, c->link_lo, c->link_hi);
        return;
    }
    auto range = [](const std::vector<char> &v, char what, i64 &lo, i64 &hi) {
        lo = hi = 0; bool any = false;
        for (size_t i = 0; i < v.size(); ++i) if (v[i] == what) { if (!any) { lo = (i64)i; any = true; } hi = (i64)i + 1; }
    };
    range(S.col_local, 1, c->col_lo, c->col_hi);
    range(S.row_local, 1, c->row_lo, c->row_hi);
    range(S.row_local, 2, c->link_lo, c->link_hi);
}

}  // namespace

// ---- the two KKT calls of the device-resident interior-point loops on a multi-device handle (tlpk_ipm.cpp) ----
extern "C++" {
int multi_update_resident(tlpk_handle *h) {
    const double t_in = now_ms();
    if (int rc = for_shards(h, [](tlpk_handle *c, int) { return tlpk_update_local(c, c->d_theta, c->d_regP, c->d_regD); })) return rc;
    return multi_update_tail(h, t_in);
}
int multi_solve_resident(tlpk_handle *h, double *const *dx, double *const *dy, const double *const *xip, const double *const *xid) {
    if (!h->factored) return TLPK_NOT_FACTORED;
    if (int rc = for_shards(h, [&](tlpk_handle *c, int r) {
            c->rhs_all_ranks = true;                     // every shard's xi_p counts on the linking rows (partial residuals)
            const int q = tlpk_solve_local(c, xip[r], xid[r]);
            c->rhs_all_ranks = false;
            return q;
        })) return rc;
    if (int rc = multi_allreduce(h, 1)) return rc;
    if (int rc = for_shards(h, [&](tlpk_handle *c, int r) {
            c->shared_dy = nullptr; c->dx_local_only = false;              // the solution stays shard-resident
            return tlpk_solve_finish(c, dx[r], dy[r], xid[r]);
        })) return rc;
    // iterative refinement (Backend(ngpus = N, refine = k) under the device-resident loops; round-3 advisor finding: only the host-pointer
    // tlpk_solve refined): one more split solve per step on the residuals every shard forms for the rows / columns it owns -- with the SAME
    // convention as the solve above, every shard's xi_p counting on the linking rows
    if (h->refine_steps > 0) {
        double norm[2] = {0.0, 0.0}; bool stop = false;
        h->refine_rejected = 0;
        if (int rc = multi_resid_norm(h, dx, dy, xip, xid, true, norm)) return rc;
        for (int it = 0; it < h->refine_steps && !stop; ++it) {
            if (int rc = multi_guarded_step(h, dx, dy, xip, xid, true, norm, &stop, [&]() -> int {
                    if (int q = for_shards(h, [&](tlpk_handle *c, int r) {
                            c->rhs_all_ranks = true;
                            const int q2 = tlpk_refine_local(c, dx[r], dy[r], xip[r], xid[r]);
                            c->rhs_all_ranks = false;
                            return q2;
                        })) return q;
                    if (int q = multi_allreduce(h, 1)) return q;
                    return for_shards(h, [&](tlpk_handle *c, int r) { return tlpk_refine_finish(c, dx[r], dy[r]); });
                })) return rc;
        }
    }
    return TLPK_OK;
}
// the pair of solves of an HSD iteration (h-system + predictor) in one pass over every shard's factor
int multi_solve2_resident(tlpk_handle *h, double *const *dx0, double *const *dy0, const double *const *xip0, const double *const *xid0,
                          double *const *dx1, double *const *dy1, const double *const *xip1, const double *const *xid1) {
    if (!h->factored) return TLPK_NOT_FACTORED;
    // refinement, or the launch-per-block schedule (TLPK_SWEEP=0, which has no two-rhs kernels): two ordinary solves, as tlpk_solve2_device
    // does on one device (round-3 advisor finding: the pair aborted here with BADARG while the single-device loop worked)
    if (h->refine_steps > 0 || !h->sub[0]->S.sweep) {
        if (int rc = multi_solve_resident(h, dx0, dy0, xip0, xid0)) return rc;
        return multi_solve_resident(h, dx1, dy1, xip1, xid1);
    }
    if (int rc = for_shards(h, [&](tlpk_handle *c, int r) {
            c->rhs_all_ranks = true;
            const int q = tlpk_solve2_local(c, xip0[r], xid0[r], xip1[r], xid1[r]);
            c->rhs_all_ranks = false;
            return q;
        })) return rc;
    if (int rc = multi_allreduce(h, 1)) return rc;
    if (int rc = multi_allreduce(h, 2)) return rc;
    return for_shards(h, [&](tlpk_handle *c, int r) { return tlpk_solve2_finish(c, dx0[r], dy0[r], xid0[r], dx1[r], dy1[r], xid1[r]); });
}
}  // extern "C++"

int tlpk_create_multi(tlpk_handle **out, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, const double *nzval,
                      int index_base, const tlpk_options *uopt, int ngpus, const int32_t *devices) {
    if (!out) return TLPK_BADARG;
    *out = nullptr;
    if (!uopt || uopt->struct_size != (int32_t)sizeof(tlpk_options) || ngpus < 1 || ngpus > MAX_DEVICES ||
        (!uopt->row_block && !uopt->detect_blocks))
        return TLPK_BADARG;                              // block-angular LPs only (general sparse LPs stay single-GPU)
    tlpk_handle *h = new (std::nothrow) tlpk_handle();
    if (!h) return TLPK_OOM;
    int rc = TLPK_OK;
    try {
        tlpk_options base = *uopt;
        if (!base.row_block) {                           // detect once, every shard gets the same explicit map
            h->row_block_copy.assign((size_t)std::max<int64_t>(m, 1), 0);
            int64_t nb = 1;
            rc = tlpk_detect_blocks(m, n, colptr, rowval, index_base, base.max_link_rows, h->row_block_copy.data(), &nb, nullptr);
            if (rc == TLPK_OK && nb < 2) { rc = TLPK_BADARG; h->last_error = "no block-angular structure found"; }
            base.row_block = h->row_block_copy.data(); base.detect_blocks = 0;
        }
        // ONE host analyse for the job: ordering, elimination tree, supernodes and front structures do not depend on the rank;
        // each shard then adds its ownership, storage offsets, lists and schedules (in parallel on the host), and uploads
        Symbolic common;
        const auto t0 = std::chrono::steady_clock::now();
        if (rc == TLPK_OK) {
            Options o; o.ordering = base.ordering; o.relax = base.relax; o.nranks = ngpus; o.row_block = base.row_block;
            o.analyse_div = 1;                          // ONE analysis for the whole job runs here: all of the host's analyse threads (the shards' rank parts, in parallel, divide by ngpus)
            rc = (base.system == TLPK_SYSTEM_K2) ? analyse_k2_common(common, m, n, colptr, rowval, nzval, index_base, o, nullptr)
                                                 : analyse_common(common, m, n, colptr, rowval, nzval, index_base, o);
            if (rc != TLPK_OK) h->last_error = common.error;
        }
        // iterative refinement is driven from here (every step holds a reduction across the shards): the shards themselves are
        // created without it
        h->refine_steps = base.refine_steps;
        if (base.refine_steps < 0 || (base.refine_steps > 0 && base.system == TLPK_SYSTEM_K2)) {
            if (rc == TLPK_OK) { rc = TLPK_BADARG; h->last_error = "refine_steps: K1 only, >= 0"; }
        }
        base.refine_steps = 0;
        std::vector<int> rcs((size_t)ngpus, TLPK_OK);
        std::vector<tlpk_options> opts((size_t)ngpus, base);
        if (rc == TLPK_OK) {
            for (int r = 0; r < ngpus; ++r) {
                tlpk_handle *c = new (std::nothrow) tlpk_handle();
                if (!c) { rc = TLPK_OOM; break; }
                h->sub.push_back(c);
                opts[(size_t)r].device = devices ? devices[r] : r;
                opts[(size_t)r].rank = r; opts[(size_t)r].nranks = ngpus;
                if (devices) for (int q = 0; q < ngpus; ++q) if (q != r && devices[q] == devices[r]) c->shared_device = true;
            }
        }
        if (rc == TLPK_OK) {
            std::vector<std::thread> pool;
            auto work = [&](int r) {
                try { rcs[(size_t)r] = create_host(h->sub[(size_t)r], opts[(size_t)r], m, n, colptr, rowval, nzval, index_base, &common); }
                catch (...) { rcs[(size_t)r] = TLPK_OOM; }
            };
            try { for (int r = 1; r < ngpus; ++r) pool.emplace_back(work, r); } catch (...) { /* fewer threads: the rest runs below */ }
            work(0);
            for (auto &t : pool) t.join();
            for (int r = (int)pool.size() + 1; r < ngpus; ++r) work(r);
            for (int r = 0; r < ngpus && rc == TLPK_OK; ++r) if (rcs[(size_t)r] != TLPK_OK) { rc = rcs[(size_t)r]; h->last_error = h->sub[(size_t)r]->last_error; }
        }
        h->ms_analyse = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (int r = 0; r < ngpus && rc == TLPK_OK; ++r) {
            rc = create_device(h->sub[(size_t)r], opts[(size_t)r]);
            if (rc != TLPK_OK) h->last_error = h->sub[(size_t)r]->last_error;
            else shard_ranges(h->sub[(size_t)r]);
        }
    } catch (...) { rc = TLPK_OOM; }
    if (rc == TLPK_OK) {
        tlpk_handle *lead = h->sub[0];
        h->S.m = m; h->S.n = n; h->has_device = true; h->device = lead->device; h->opt.nranks = 1;
        if (lead->S.system == 1) { h->S.system = 1; h->S.k2_n = n; h->S.k2_m = m; }     // user dimensions / KKT.linear_system of the job
        double *p = nullptr; int64_t cnt = 0;
        tlpk_root_panel(lead, &p, &cnt);
        const i64 tmp_len = (i64)ngpus * cnt + m;                            // staging of the peers' root buffers, the reduced buffer, the lead's rank-local dy
        h->multi_red_off = (i64)(ngpus - 1) * cnt;
        h->multi_dy0_off = (i64)ngpus * cnt;
        hipError_t e = hipSetDevice(lead->device);
        if (e == hipSuccess) e = hipMalloc((void **)&h->multi_tmp, (size_t)std::max<i64>(tmp_len, 1) * 8);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->multi_done, hipEventDisableTiming);
        {
            // TLPK_MULTI_REDUCE = rs (default: reduce-scatter + all-gather over peer copies) | gather (round 2/3: gather to the lead, ordered sum, broadcast) | rccl
            const char *mr = std::getenv("TLPK_MULTI_REDUCE");
            h->multi_mode = (mr && std::string(mr) == "gather") ? 0 : 1;
            h->rs_slice = ((cnt + ngpus - 1) / ngpus + 15) / 16 * 16;
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->multi_ev[0], hipEventDisableTiming);
        for (int r = 0; r < ngpus && e == hipSuccess; ++r) {
            e = hipSetDevice(h->sub[r]->device);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&h->multi_ev2[r], hipEventDisableTiming);
            if (e == hipSuccess && ngpus > 1) {
                void *q = nullptr;
                e = hipMalloc(&q, (size_t)std::max<i64>((i64)(ngpus - 1) * h->rs_slice, 1) * 8);
                if (e == hipSuccess) { h->sub[r]->rs_stage = (double *)q; h->sub[r]->allocs.push_back(q); }
            }
            // reduce-scatter + all-gather: every shard copies into every other shard's staging / root buffers.  hipMemcpyPeerAsync needs no peer
            // mapping, but a mapped peer lets the copy engine go straight over the link; a pair that cannot be mapped (a node without full
            // connectivity) is NOT an error: the reduction falls back to gather-to-lead, which only needs the rank-to-lead mappings checked
            // below (round-4 advisor finding: the all-pairs requirement made creates fail that worked in round 3).
            for (int t = 0; t < ngpus && e == hipSuccess && h->multi_mode == 1; ++t)
                if (h->sub[t]->device != h->sub[r]->device) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(h->sub[t]->device, 0);
                    (void)hipGetLastError();
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) h->multi_mode = 0;
                }
        }
        for (int r = 1; r < ngpus && e == hipSuccess; ++r) {
            e = hipSetDevice(h->sub[r]->device);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&h->multi_ev[r], hipEventDisableTiming);
            if (e == hipSuccess && h->sub[r]->device != lead->device) {
                const hipError_t pe = hipDeviceEnablePeerAccess(lead->device, 0);      // this rank stores into the lead's dx / dy
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) e = pe;
                (void)hipGetLastError();
            }
        }
        if (e != hipSuccess) rc = hip_fail(h, e, "multi-device init");
    }
    if (rc == TLPK_OK) {
        // library-owned reductions over RCCL (north_star: "RCCL-reducing the linking-block Schur complement over xGMI"): opt-in,
        // needs one DISTINCT device per shard (RCCL refuses two ranks on one device)
        const char *e = std::getenv("TLPK_MULTI_REDUCE");
        if (e && std::string(e) == "rccl" && ngpus > 1) {
            bool distinct = true;
            for (int a = 0; a < ngpus; ++a) for (int b = a + 1; b < ngpus; ++b) if (h->sub[a]->device == h->sub[b]->device) distinct = false;
            std::string err;
            if (!distinct) { h->last_error = "TLPK_MULTI_REDUCE=rccl needs one distinct device per shard"; rc = TLPK_BADARG; }
            else if (!g_rccl.load(err)) { h->last_error = err; rc = TLPK_HIPERR; }
            else {
                int devs[MAX_DEVICES];
                for (int r = 0; r < ngpus; ++r) devs[r] = h->sub[r]->device;
                const int q = g_rccl.CommInitAll(h->multi_comm, ngpus, devs);
                if (q != 0) { h->last_error = std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(q) : "error"); rc = TLPK_HIPERR; }
                else h->multi_rccl = true;
            }
        }
    }
    if (rc != TLPK_OK) { g_create_error = h->last_error.empty() ? std::string(tlpk_strerror(rc)) : h->last_error; tlpk_destroy(h); return rc; }     // tlpk_last_create_error()
    g_create_error.clear();
    *out = h;
    return TLPK_OK;
}

// ---- introspection ----
int tlpk_info(const tlpk_handle *h, tlpk_stats *out) {
    if (!h || !out) return TLPK_BADARG;
    if (!h->sub.empty()) {                                  // multi-device handle: the lead's view of the job
        const int rc = tlpk_info(h->sub[0], out);
        out->fail_col = h->fail_col;
        i64 bytes = 0; i32 nloc = 0;
        for (const tlpk_handle *c : h->sub) { bytes += c->device_bytes; nloc += c->S.n_local_blocks; }
        out->device_bytes = bytes; out->n_local_blocks = nloc;
        out->ms_analyse = h->ms_analyse; out->ms_last_update = h->ms_update; out->ms_enqueue_update = h->ms_enqueue_update;
        out->refine_rejected = h->refine_rejected;
        return rc;
    }
    const Symbolic &S = h->S;
    std::memset(out, 0, sizeof(*out));
    out->m = user_m(h); out->n = user_n(h); out->nnzA = (S.system == 1) ? S.n : S.nnzA; out->nnzS = S.nnzS; out->nnzL = S.nnzL;
    out->nnzL_stored = S.lval_len; out->flops_chol = S.flops_chol; out->flops_panel = S.flops_panel;
    out->n_supernodes = S.nsuper; out->n_levels = S.nlevels; out->max_front = S.max_front;
    out->n_pairs = S.pair_ptr.empty() ? 0 : S.pair_ptr.back();
    out->device_bytes = h->device_bytes;
    for (const Launch &L : S.factor_launches) if (L.kind != LK_ALLREDUCE_ROOT) out->launches_update++;
    out->launches_update += 3;
    for (const Launch &L : S.fwd_launches) if (L.kind != LK_ALLREDUCE_ROOT) out->launches_solve++;
    out->launches_solve += (i64)S.bwd_launches.size() + 3;
    out->fail_col = h->fail_col;
    out->ms_analyse = h->ms_analyse; out->ms_last_update = h->ms_update; out->ms_last_solve = h->ms_solve;
    out->n_local_blocks = S.n_local_blocks; out->n_blocks = S.nblocks;
    out->flops_update = S.flops_update;
    out->flops_update_alg = S.flops_update_alg;
    out->flops_update_chain = S.flops_update_chain; out->flops_update_alg_chain = S.flops_update_alg_chain;
    for (const Launch &L : S.factor_launches) if (L.kind == LK_CHAIN) { out->chain_launches++; out->chain_items += L.count; }
    out->refine_rejected = h->refine_rejected;
    out->root_panel_len = (S.root_front >= 0) ? pk_len(S.fronts[S.root_front].lda, S.fronts[S.root_front].ns) : 0;
    return TLPK_OK;
}

int tlpk_kernel_timing(const tlpk_handle *h, tlpk_kernel_times *out) {
    if (!h || !out) return TLPK_BADARG;
    *out = h->kt;
    return TLPK_OK;
}

int tlpk_set_profile(tlpk_handle *h, int on) {
    if (!h) return TLPK_BADARG;
    h->profile = on != 0;
    h->ev_used = 0; h->ev_class.clear(); h->ev_launch.clear();
    std::memset(&h->kt, 0, sizeof(h->kt));
    return TLPK_OK;
}

int tlpk_get_perm(const tlpk_handle *h, int64_t *perm) {
    if (!h || !perm) return TLPK_BADARG;
    if (!h->sub.empty()) return tlpk_get_perm(h->sub[0], perm);
    for (i64 i = 0; i < h->S.m; ++i) perm[i] = h->S.perm[(size_t)i];
    return TLPK_OK;
}

int64_t tlpk_symbolic_get(const tlpk_handle *h, const char *what, int64_t *buf, int64_t cap) {
    if (!h || !what) return -1;
    if (!h->sub.empty() && std::string(what) != "chain_retries") return tlpk_symbolic_get(h->sub[0], what, buf, cap);
    const Symbolic &S = h->S;
    std::vector<i64> tmp;
    const std::string w(what);
    auto from32 = [&](const auto &v) { tmp.assign(v.begin(), v.end()); };
    auto field = [&](auto getter) { tmp.resize(S.fronts.size()); for (size_t s = 0; s < S.fronts.size(); ++s) tmp[s] = (i64)getter(S.fronts[s]); };
    if (w == "perm") from32(S.perm);
    else if (w == "etree") from32(S.parent);
    else if (w == "colcount") from32(S.colcount);
    else if (w == "s_colptr") tmp = S.Sp;
    else if (w == "s_rowidx") from32(S.Si);
    else if (w == "s_target") from32(S.s_target);
    else if (w == "s_diag_row") from32(S.s_diag_row);
    else if (w == "pair_ptr") from32(S.pair_ptr);
    else if (w == "pair_j") from32(S.pair_j);
    else if (w == "rowidx") from32(S.rowidx);
    else if (w == "rel") from32(S.rel);
    else if (w == "ea_tab") from32(S.ea_tab);
    else if (w == "front_eatab") field([](const FrontDesc &f) { return f.eatab; });
    else if (w == "children") from32(S.children);
    else if (w == "depth") from32(S.depth);
    else if (w == "front_block") from32(S.front_block);
    else if (w == "row_block") tmp = h->row_block_copy;             // the block map in use (given or detected); empty = general sparse
    else if (w == "front_group") from32(S.front_group);
    else if (w == "ngroups") tmp.assign(1, S.ngroups);
    else if (w == "front_local") tmp.assign(S.front_local.begin(), S.front_local.end());
    else if (w == "col_local") tmp.assign(S.col_local.begin(), S.col_local.end());
    else if (w == "row_local") tmp.assign(S.row_local.begin(), S.row_local.end());
    else if (w == "front_f") field([](const FrontDesc &f) { return f.f; });
    else if (w == "front_lda") field([](const FrontDesc &f) { return f.lda; });
    else if (w == "front_ns") field([](const FrontDesc &f) { return f.ns; });
    else if (w == "front_col0") field([](const FrontDesc &f) { return f.col0; });
    else if (w == "front_parent") field([](const FrontDesc &f) { return f.parent; });
    else if (w == "front_loff") field([](const FrontDesc &f) { return f.loff; });
    else if (w == "front_rowoff") field([](const FrontDesc &f) { return f.rowoff; });
    else if (w == "front_reloff") field([](const FrontDesc &f) { return f.reloff; });
    else if (w == "front_child_ptr") field([](const FrontDesc &f) { return f.child_ptr; });
    else if (w == "front_nchild") field([](const FrontDesc &f) { return f.nchild; });
    else if (w == "root_front") tmp.assign(1, S.root_front);
    else if (w == "potrf_tasks") { for (auto &t : S.potrf_tasks) { tmp.push_back(t.front); tmp.push_back(t.k0); tmp.push_back(t.nb); tmp.push_back(t.kprev); } }
    else if (w == "trsm_tasks") { for (auto &t : S.trsm_tasks) { tmp.push_back(t.front); tmp.push_back(t.k0); tmp.push_back(t.nb); tmp.push_back(t.row0); tmp.push_back(t.kprev); tmp.push_back(t.pad1); } }
    else if (w == "update_tasks") { for (auto &t : S.update_tasks) { tmp.push_back(t.front); tmp.push_back(t.k0); tmp.push_back(t.kw); tmp.push_back(t.i0); tmp.push_back(t.j0); tmp.push_back(t.jlim); tmp.push_back(t.beta0); tmp.push_back(t.pad1); tmp.push_back(t.seg); tmp.push_back(t.nsl); } }
    else if (w == "upd_seg") from32(S.upd_seg);
    else if (w == "trsm_early") { for (auto &t : S.trsm_tasks) tmp.push_back(t.pad2); }        // > 0: early entry, the counter (global index + 1) of the block column's diagonal block
    else if (w == "update_tile64") { for (auto &t : S.update_tasks) tmp.push_back(t.pad2); }      // 1 = a 64 x 64 tile (chain launches: the diagonal block's short update)
    else if (w == "skip_off") tmp = S.skip_off;
    else if (w == "skip_bits") { tmp.resize(S.skip_bits.size()); std::memcpy(tmp.data(), S.skip_bits.data(), S.skip_bits.size() * 8); }
    else if (w == "front_single") tmp.assign(S.front_single.begin(), S.front_single.end());
    else if (w == "reduce_tasks") { for (auto &t : S.reduce_tasks) { tmp.push_back(t.front); tmp.push_back(t.k0); tmp.push_back(t.kw); tmp.push_back(t.i0); tmp.push_back(t.j0); tmp.push_back(t.jlim); tmp.push_back(t.beta0); tmp.push_back(t.pad1); } }
    else if (w == "chain_items") { for (auto &t : S.chain_items) { for (i32 v : {t.role, t.task, t.sub, t.w0, t.n0, t.need0, t.w1, t.n1, t.need1, t.w2, t.need2, t.sig}) tmp.push_back(v); } }
    else if (w == "chain_counters") tmp.assign(1, S.chain_counters);
    else if (w == "chain_retries") tmp.assign(1, h->chain_retries);           // updates of this handle that were replayed after a dependency-driven launch gave up waiting (tlpk_update)
    else if (w == "chain_trace") {                               // diagnostics: the time stamps of the last update (the caller has synchronised)
        if (!h->d.chain_trace) return -1;
        tmp.resize(4 * S.chain_items.size());
        if (hipMemcpy(tmp.data(), h->d.chain_trace, tmp.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    else if (w == "fa_tasks") { for (auto &t : S.fa_tasks) { tmp.push_back(t.front); tmp.push_back(t.bc); tmp.push_back(t.br0); tmp.push_back(t.br1); } }
    else if (w == "front_fa") tmp.assign(S.front_fa.begin(), S.front_fa.end());
    else if (w == "front_upper") tmp.assign(S.front_upper.begin(), S.front_upper.end());
    else if (w == "ea_tasks") { for (auto &t : S.ea_tasks) { tmp.push_back(t.front); tmp.push_back(t.j0); tmp.push_back(t.j1); tmp.push_back(t.bidx); tmp.push_back(t.br0); tmp.push_back(t.br1); } }
    else if (w == "fwd_gather_tasks" || w == "fwd_diag_tasks" || w == "fwd_update_tasks" || w == "bwd_update_tasks" || w == "fwd_small_tasks" || w == "bwd_small_tasks" ||
             w == "fwd_sweep_tasks" || w == "bwd_sweep_tasks") {
        const std::vector<SolveTask> &v = (w == "fwd_gather_tasks") ? S.fwd_gather_tasks : (w == "fwd_diag_tasks") ? S.fwd_diag_tasks :
                                          (w == "fwd_update_tasks") ? S.fwd_update_tasks : (w == "bwd_update_tasks") ? S.bwd_update_tasks :
                                          (w == "fwd_small_tasks") ? S.fwd_small_tasks : (w == "bwd_small_tasks") ? S.bwd_small_tasks :
                                          (w == "fwd_sweep_tasks") ? S.fwd_sweep_tasks : S.bwd_sweep_tasks;
        for (auto &t : v) { tmp.push_back(t.front); tmp.push_back(t.k0); tmp.push_back(t.nb); tmp.push_back(t.row0); tmp.push_back(t.slot); tmp.push_back(t.nslot); }
    }
    else if (w == "front_flagoff") field([](const FrontDesc &f) { return f.flagoff; });
    else if (w == "n_sweep_flags") tmp.assign(1, S.n_sweep_flags);
    else if (w == "front_ucoff") field([](const FrontDesc &f) { return f.ucoff; });
    else if (w == "front_uoff") field([](const FrontDesc &f) { return f.uoff; });
    else if (w == "front_ubuf") field([](const FrontDesc &f) { return f.ubuf; });
    else if (w == "gth_ptr") tmp.assign(S.gth_ptr.begin(), S.gth_ptr.end());
    else if (w == "gth_src") tmp.assign(S.gth_src.begin(), S.gth_src.end());
    else if (w == "factor_launches") { for (auto &L : S.factor_launches) { tmp.push_back(L.kind); tmp.push_back(L.first); tmp.push_back(L.count); } }
    else if (w == "fwd_launches") { for (auto &L : S.fwd_launches) { tmp.push_back(L.kind); tmp.push_back(L.first); tmp.push_back(L.count); } }
    else if (w == "bwd_launches") { for (auto &L : S.bwd_launches) { tmp.push_back(L.kind); tmp.push_back(L.first); tmp.push_back(L.count); } }
    else return -1;
    const i64 len = (i64)tmp.size();
    if (buf) std::copy(tmp.begin(), tmp.begin() + std::min(len, cap), buf);
    return len;
}

int64_t tlpk_symbolic_get_f64(const tlpk_handle *h, const char *what, double *buf, int64_t cap) {
    if (!h || !what) return -1;
    const std::string w(what);
    if (w != "pair_w") return -1;
    const auto &v = h->S.pair_w;
    const i64 len = (i64)v.size();
    if (buf) std::copy(v.begin(), v.begin() + std::min(len, cap), buf);
    return len;
}

int tlpk_get_factor(tlpk_handle *h, double *lval, int64_t cap) {
    if (!h || !lval) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "multi-device handle: only tlpk_update / tlpk_solve / tlpk_info / tlpk_destroy apply"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (cap < h->S.lval_len) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(lval, h->d.ctx.Lval, (size_t)h->S.lval_len * 8, hipMemcpyDeviceToHost));
    return TLPK_OK;
}

const char *tlpk_strerror(int code) {
    switch (code) {
    case TLPK_OK: return "ok";
    case TLPK_NOT_POSDEF: return "matrix is not positive definite";
    case TLPK_BADARG: return "bad argument / dimension mismatch";
    case TLPK_OOM: return "out of memory";
    case TLPK_HIPERR: return "HIP runtime error";
    case TLPK_NO_DEVICE: return "no HIP device (analyse-only handle or no GPU visible)";
    case TLPK_TOO_LARGE: return "factor does not fit the memory budget";
    case TLPK_NOT_FACTORED: return "solve called before a successful update";
    case TLPK_INTERNAL: return "internal error";
    default: return "unknown error";
    }
}
const char *tlpk_last_error(const tlpk_handle *h) { return h ? h->last_error.c_str() : ""; }
const char *tlpk_backend_name(void) { return "HIP (gfx950)"; }
const char *tlpk_system_name(void) { return "Normal equations (K1)"; }
const char *tlpk_linear_system(const tlpk_handle *h) { return (h && h->S.system == 1) ? "Augmented system (K2)" : "Normal equations (K1)"; }

}  // extern "C"
