// symbolic.cpp -- host analyse phase: pattern of S = A*A', fill-reducing ordering, elimination
// tree, column counts, supernodes (fronts), relative indices, assembly lists for A*D*A' + Rd.
//
// Reference counterpart: the `cholesky(Symmetric(A*A' + I))` call in KKT.setup
// (/root/reference/src/KKT/Cholmod/spd.jl:14-17), i.e. CHOLMOD's analyse [ext].  The pattern of S
// is structural and constant over the IPM run (spd.jl:14,43), so everything computed here is
// reused by every update!/solve!.
#include "tlpk_host.hpp"
#include "../../include/tlpk.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdio>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <mutex>
#include <functional>
#include <condition_variable>
#include <map>
#include <climits>

namespace tlpk {

// Extend-add ranges of a parent front (one workgroup each): ea_cols(p) columns wide, counted from 0 inside the pivot
// columns [0, ns) and from ns inside the update-matrix columns [ns, f).  Boundary k of ea_nbounds(p):
//   k < npan: k * cols ; k == npan: ns ; k > npan: ns + (k - npan) * cols, the last one being f.
static inline i32 ea_cols_big() {              // TLPK_EA_COLS (tuning knob, 4 .. EA_COLS): parent columns per extend-add workgroup of the big fronts
    static const i32 v = [] { const char *e = std::getenv("TLPK_EA_COLS"); const int c = e ? std::atoi(e) : EA_COLS; return (i32)std::max(4, std::min(c, EA_COLS)); }();
    return v;
}
// Front assembly (k_front_assemble, FaTask) -- an experiment of round 4, OFF by default (TLPK_FA_MIN_F=512 turns it on): the panel of a front
// is FORMED tile by tile in LDS (S entries + children in child order, written once) instead of zero-fill + k_assemble + read-modify-write
// extend-add.  Its extend-add ranges are FA_CW = 16 columns wide, the same boundaries cut the rows of a tile.  Chosen per front in analyse_rank
// (Symbolic::front_fa): >= fa_min_f() rows and on average >= fa_density() contributions per entry of the front.
// Measured (profiles/r04_front_assembly.txt): on every large front (TLPK_FA_DENSITY=0) config C4 extend-add + assembly 6.9 -> 9.0 ms, north-star
// instance 12.4 -> 27.4 ms -- a tile sees ~100..400 children that each bring a few dozen entries, i.e. two dependent memory round trips per
// (child, tile) with nothing to overlap them, where the column-oriented k_extend_add streams whole child columns; on the linking (root) front only
// (one dense update matrix per diagonal block, the default density threshold): 7.16 vs 7.05 ms, no gain.  Parity-green (also on NaN-poisoned storage).
static inline i32 fa_min_f() {
    static const i32 v = [] { const char *e = std::getenv("TLPK_FA_MIN_F"); const int c = e ? std::atoi(e) : 0; return (i32)(c <= 0 ? INT32_MAX : c); }();
    return v;
}
static inline double fa_density() {
    static const double v = [] { const char *e = std::getenv("TLPK_FA_DENSITY"); return e ? std::atof(e) : 4.0; }();
    return v;
}
static inline i32 ea_cols(const FrontDesc &p, bool fa) { return fa ? FA_CW : ((p.f >= 2048) ? ea_cols_big() : 4); }     // small fronts: more, narrower workgroups
static inline i32 ea_npan(const FrontDesc &p, bool fa) { const i32 c = ea_cols(p, fa); return (p.ns + c - 1) / c; }
static inline i32 ea_nbounds(const FrontDesc &p, bool fa) { const i32 c = ea_cols(p, fa); return ea_npan(p, fa) + (p.f - p.ns + c - 1) / c + 1; }
static inline i32 ea_bound(const FrontDesc &p, bool fa, i32 k) {
    const i32 c = ea_cols(p, fa), npan = ea_npan(p, fa);
    return (k < npan) ? k * c : std::min(p.f, p.ns + (k - npan) * c);
}


namespace {

// Liu's elimination-tree algorithm with path compression on a graph given in ORIGINAL labels
// plus a labelling iperm (old -> new); returns parent in NEW labels.
// Liu's algorithm with path compression for the nodes [i0, i1) of the ordering.  `parent` and `anc`
// (both initialised to -1 by the caller) are only touched at positions < i1 that are connected to
// the range, so disjoint ranges of mutually non-adjacent node sets can run concurrently.
static void etree_range(i32 i0, i32 i1, const std::vector<i64> &xadj, const std::vector<i32> &adj, const std::vector<i32> &perm,
                        const std::vector<i32> &iperm, std::vector<i32> &parent, std::vector<i32> &anc) {
    for (i32 i = i0; i < i1; ++i) {
        const i32 old = perm[i];
        for (i64 p = xadj[old]; p < xadj[old + 1]; ++p) {
            i32 k = iperm[adj[p]];
            while (k != -1 && k < i) {
                const i32 nxt = anc[k];
                anc[k] = i;
                if (nxt == -1) parent[k] = i;
                k = nxt;
            }
        }
    }
}
void etree_of(i32 m, const std::vector<i64> &xadj, const std::vector<i32> &adj, const std::vector<i32> &perm,
              const std::vector<i32> &iperm, std::vector<i32> &parent) {
    parent.assign(m, -1);
    std::vector<i32> anc(m, -1);
    etree_range(0, m, xadj, adj, perm, iperm, parent, anc);
}

// Postorder of a forest (children visited in increasing label order).  `skip[v]` nodes are cut
// out of the forest (their children become roots) and are not emitted.
void postorder_forest(i32 m, const std::vector<i32> &parent, const std::vector<char> *skip, std::vector<i32> &post) {
    std::vector<i32> head(m, -1), next(m, -1), roots;
    for (i32 v = m - 1; v >= 0; --v) {
        if (skip && (*skip)[v]) continue;
        const i32 p = parent[v];
        if (p == -1 || (skip && (*skip)[p])) continue;
        next[v] = head[p];
        head[p] = v;
    }
    post.clear();
    post.reserve(m);
    std::vector<i32> stack;
    for (i32 r = 0; r < m; ++r) {
        if (skip && (*skip)[r]) continue;
        const i32 p = parent[r];
        if (!(p == -1 || (skip && (*skip)[p]))) continue;
        stack.push_back(r);
        while (!stack.empty()) {
            const i32 v = stack.back();
            const i32 c = head[v];
            if (c != -1) { head[v] = next[c]; stack.push_back(c); }
            else { post.push_back(v); stack.pop_back(); }
        }
    }
}

}  // namespace

static int fail(Symbolic &S, int code, const std::string &msg) { S.error = msg; return code; }

static void build_schedule(Symbolic &S);

// Host threads for the embarrassingly parallel parts of the analyse phase.  fn(thread, i) is called once
// for every i in [0, n), items handed out dynamically; the result never depends on the number of
// threads (every item writes its own outputs).  Returns false if a worker threw (out of memory).
// Sharded runs (round-5 review): every one of N ranks -- N processes of a multi-GPU launch, or the N per-shard threads of tlpk_create_multi -- runs this
// analysis at the same time on the same host: the default thread count is divided by N (8 ranks would otherwise start 512 analyse threads on a host that
// grants the job 16 - 256 CPUs).  Set at the top of analyse_common / analyse_rank from Options.nranks; an explicit TLPK_HOST_THREADS is taken as given.
static thread_local i64 g_host_thread_div = 1;
static unsigned host_threads(i64 n) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    // up to a quarter of the hardware threads, at most 64 (env TLPK_HOST_THREADS overrides): the per-front / per-block phases
    // of a block-angular LP with 64 diagonal blocks were 3-4 rounds deep with the old cap of 16 on a 256-thread host.  NOT capped by the
    // container's CPU quota (16 CPUs on the GPU boxes): the phases are bursts of well under the 100-ms accounting period, 64 threads are the
    // fastest setting there (C4 281 ms against 347 with 16; north-star LP 1093 against 1258: tools/analyse_threads_probe.py) -- unlike the
    // seconds-long OpenMP teams of the CPU comparator, which the quota throttles
    static const i64 cap = [] { const char *e = std::getenv("TLPK_HOST_THREADS"); return e ? std::max<i64>(1, std::atoll(e)) : (i64)64; }();
    const i64 mine = std::getenv("TLPK_HOST_THREADS") ? cap : std::max<i64>(1, std::min<i64>(cap, std::max<i64>(16, hw / 4)) / g_host_thread_div);
    return (unsigned)std::max<i64>(1, std::min<i64>({(i64)hw, mine, n}));
}
// Round 6: the worker threads of ONE analyse call.  parallel_for used to create and join its threads on every call -- ~50 calls per analyse (two per level of the
// supernodal tree alone) x up to 63 threads: tens of milliseconds of clone / join on a 250-ms analyse.  An AnalysePool lives for the duration of an analyse_common /
// analyse_rank call (scope object: nothing survives the call, nothing to make fork-safe), its threads are created on first use and sleep on a condition variable
// between jobs.  A job = (worker function, number of participants); thread t of the pool runs worker(t + 1), the caller runs worker(0) and waits for the rest.
// Nested parallel_for calls (none today) and calls outside an analyse fall back to the create-and-join form.
struct AnalysePool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    const std::function<void(unsigned)> *job = nullptr;
    unsigned want = 0, done = 0;
    unsigned long long gen = 0;
    bool stop = false, busy = false;
    ~AnalysePool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    void loop(unsigned idx) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *j = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                if (idx < want) j = job;
            }
            if (j) {
                (*j)(idx + 1);
                std::lock_guard<std::mutex> lk(mu);
                if (++done == want) cv_done.notify_one();
            }
        }
    }
    // runs worker(0 .. nthreads - 1); false = the pool cannot serve (busy: a nested call) and the caller must use its own threads
    bool run(unsigned nthreads, const std::function<void(unsigned)> &worker) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (busy) return false;
            busy = true;
        }
        unsigned helpers = nthreads - 1;
        try { while (th.size() < helpers) { const unsigned idx = (unsigned)th.size(); th.emplace_back([this, idx] { loop(idx); }); } }
        catch (...) { helpers = (unsigned)th.size(); }      // thread / pid limits of a container: go on with the threads we have (the workers drain the items whoever runs them)
        {
            std::lock_guard<std::mutex> lk(mu);
            job = &worker; want = helpers; done = 0; ++gen;
        }
        cv.notify_all();
        worker(0);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return done == want; });
            job = nullptr; want = 0; busy = false;
        }
        return true;
    }
};
static thread_local AnalysePool *g_analyse_pool = nullptr;      // the pool of the analyse running on this thread (TLPK_ANALYSE_POOL=0: none)
struct AnalysePoolScope {
    AnalysePool pool; AnalysePool *prev;
    AnalysePoolScope() : prev(g_analyse_pool) {
        static const bool on = [] { const char *e = std::getenv("TLPK_ANALYSE_POOL"); return !e || std::atoi(e) != 0; }();
        if (on && !prev) g_analyse_pool = &pool;
    }
    ~AnalysePoolScope() { g_analyse_pool = prev; }
};

// `chunk` consecutive items go to the same thread (neighbouring items usually write neighbouring memory:
// item-by-item hand-out made the threads fight over cache lines on instances with 400 000 small fronts).
template <class F>
static bool parallel_for(i64 n, unsigned nthreads, F &&fn, i64 chunk = 1) {
    std::atomic<i64> next{0};
    std::atomic<int> failed{0};
    auto worker = [&](unsigned tid) {
        try {
            for (i64 i0; (i0 = next.fetch_add(chunk)) < n;)
                for (i64 i = i0; i < std::min(n, i0 + chunk); ++i) fn(tid, i);
        } catch (...) { failed = 1; }
    };
    if (nthreads <= 1) { worker(0); return !failed; }
    if (g_analyse_pool) {
        const std::function<void(unsigned)> w = worker;
        if (g_analyse_pool->run(nthreads, w)) return !failed;
    }
    // A thread that cannot be created (thread / pid limits of a container) is not an error: the
    // threads that did start and the calling thread drain the work.  Nothing may escape while a
    // started thread is still joinable (the vector's destructor would call std::terminate).
    std::vector<std::thread> pool;
    try {
        pool.reserve(nthreads);
        for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker, t);
    } catch (...) { /* std::system_error / bad_alloc: go on with the threads we have */ }
    worker(0);
    for (auto &th : pool) th.join();
    return !failed;
}

// Same, for call sites without an error path of their own: a failed worker (out of memory) becomes a
// std::bad_alloc AFTER every thread has been joined; tlpk_create maps it to TLPK_OOM.
template <class F>
static void parallel_for_throw(i64 n, unsigned nthreads, F &&fn, i64 chunk = 1) {
    if (!parallel_for(n, nthreads, std::forward<F>(fn), chunk)) throw std::bad_alloc();
}

// independent iterations over [0, n) in chunks of 64 K on the host threads (the element-wise relabelling loops of the analyse: 2e6 scattered 4-byte accesses each on
// the north-star LP, 5 - 10 ms apiece on one thread)
template <class F>
static void par_chunks(i64 n, F &&fn) {
    constexpr i64 CH = 65536;
    const i64 nch = (n + CH - 1) / CH;
    if (nch <= 1) { if (n > 0) fn((i64)0, n); return; }
    parallel_for_throw(nch, host_threads(nch), [&](unsigned, i64 ch) { fn(ch * CH, std::min(n, (ch + 1) * CH)); });
}

// TLPK_TIMING=1: wall time of the analyse phases on stderr
struct PhaseTimer {
    bool on; std::chrono::steady_clock::time_point t0; const char *name = nullptr;
    PhaseTimer() : on(std::getenv("TLPK_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *next) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        if (name) std::fprintf(stderr, "[tlpk analyse] %-28s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(t1 - t0).count());
        name = next; t0 = t1;
    }
};

// The analyse phase in two parts.  analyse_common: everything that does not depend on the rank of a sharded run -- copy of
// A, the graph of A*A', the ordering, the elimination tree, column counts, supernodes, amalgamation, front structures (87 % of
// the time on config C4).  analyse_rank: ownership, storage offsets, relative indices, gather / assembly lists and the launch
// schedules of ONE rank.  tlpk_create runs both; tlpk_create_multi runs the common part once and the rank part per device.
int analyse_common(Symbolic &S, i64 m64, i64 n64, const i64 *colptr, const i64 *rowval, const double *nzval,
                   int base, const Options &opt) {
    PhaseTimer pt;
    AnalysePoolScope pool_scope;
    g_host_thread_div = opt.analyse_div > 0 ? opt.analyse_div : std::max<i32>(1, opt.nranks);
    if (m64 < 0 || n64 < 0 || (base != 0 && base != 1) || !colptr) return fail(S, TLPK_BADARG, "bad dimensions or index base");
    if (m64 >= (i64)1 << 31 || n64 >= (i64)1 << 31) return fail(S, TLPK_TOO_LARGE, "m or n exceeds int32");
    const i64 nnz = colptr[n64] - base;
    if (nnz < 0 || nnz >= (i64)1 << 31) return fail(S, TLPK_TOO_LARGE, "nnz(A) exceeds int32");
    if (nnz > 0 && (!rowval || !nzval)) return fail(S, TLPK_BADARG, "null rowval/nzval");
    const i32 m = (i32)m64, n = (i32)n64;
    S.m = m; S.n = n; S.nnzA = nnz;
    if (opt.nranks < 1 || opt.rank < 0 || opt.rank >= opt.nranks) return fail(S, TLPK_BADARG, "bad rank/nranks");
    if (opt.nranks > 1 && !opt.row_block) return fail(S, TLPK_BADARG, "sharding needs row_block (general sparse LPs are single-GPU)");

    pt.mark("copy A / CSR");
    // ---- 1. copy A (CSC) and build CSR ----
    S.Ap.resize((size_t)n + 1); S.Ai.resize((size_t)nnz); S.Ax.resize((size_t)nnz); S.Acol.resize((size_t)nnz);
    for (i32 j = 0; j <= n; ++j) {
        S.Ap[j] = colptr[j] - base;
        if (S.Ap[j] < 0 || S.Ap[j] > nnz || (j > 0 && S.Ap[j] < S.Ap[j - 1])) return fail(S, TLPK_BADARG, "colptr not monotone");
    }
    // (round 5: on the host threads.  The copy by chunks of columns; the transposition by ranges of ROWS: every thread scans the row indices of the whole matrix -- a
    // sequential read of 4 nnz bytes -- and places the entries of its own rows, in column order: the same CSR arrays as the serial loop, without a shared cursor.)
    {
        constexpr i64 CH = 8192;
        const i64 nch = ((i64)n + CH - 1) / CH;
        std::atomic<int> bad{0};
        if (!parallel_for(nch, host_threads(nch), [&](unsigned, i64 ch) {
                const i32 j1 = (i32)std::min<i64>(n, (ch + 1) * CH);
                for (i32 j = (i32)(ch * CH); j < j1; ++j)
                    for (i64 p = S.Ap[j]; p < S.Ap[j + 1]; ++p) {
                        const i64 r = rowval[p] - base;
                        if (r < 0 || r >= m) { bad = 1; continue; }
                        S.Ai[p] = (i32)r; S.Ax[p] = nzval[p]; S.Acol[p] = j;
                    }
            })) return fail(S, TLPK_OOM, "out of memory in the analyse phase");
        if (bad) return fail(S, TLPK_BADARG, "row index out of range");
    }
    S.Tp.assign((size_t)m + 1, 0);
    S.Tj.resize((size_t)nnz); S.Tpos.resize((size_t)nnz);
    {
        const unsigned nt = host_threads(std::max<i64>(1, nnz / 200000));          // row ranges, one per thread
        const i64 rstep = ((i64)m + nt - 1) / nt;
        if (!parallel_for(nt, nt, [&](unsigned, i64 t) {                             // counts of the thread's rows
                const i64 r0 = t * rstep, r1 = std::min<i64>(m, r0 + rstep);
                if (r0 >= r1) return;
                for (i64 p = 0; p < nnz; ++p) { const i64 r = S.Ai[p]; if (r >= r0 && r < r1) S.Tp[r + 1]++; }
            })) return fail(S, TLPK_OOM, "out of memory in the analyse phase");
        for (i32 i = 0; i < m; ++i) S.Tp[i + 1] += S.Tp[i];
        if (!parallel_for(nt, nt, [&](unsigned, i64 t) {
                const i64 r0 = t * rstep, r1 = std::min<i64>(m, r0 + rstep);
                if (r0 >= r1) return;
                std::vector<i64> cur(S.Tp.begin() + r0, S.Tp.begin() + r1);
                for (i64 p = 0; p < nnz; ++p) {
                    const i64 r = S.Ai[p];
                    if (r < r0 || r >= r1) continue;
                    const i64 q = cur[(size_t)(r - r0)]++;
                    S.Tj[q] = S.Acol[p]; S.Tpos[q] = (i32)p;
                }
            })) return fail(S, TLPK_OOM, "out of memory in the analyse phase");
    }

    pt.mark("block structure");
    // ---- 2. block-angular structure (optional) ----
    std::vector<i32> &row_block = S.row_block_v, &col_block = S.col_block_v;
    row_block.clear(); col_block.clear();
    i32 nblocks = 0, nlink = 0;
    if (opt.row_block) {
        row_block.resize(m);
        for (i32 i = 0; i < m; ++i) {
            const i64 b = opt.row_block[i];
            if (b < -1 || b >= (i64)1 << 30) return fail(S, TLPK_BADARG, "row_block out of range");
            row_block[i] = (i32)b;
            if (b >= 0) nblocks = std::max(nblocks, (i32)b + 1); else ++nlink;
        }
        col_block.assign(n, -1);
        for (i32 j = 0; j < n; ++j)
            for (i64 p = S.Ap[j]; p < S.Ap[j + 1]; ++p) {
                const i32 b = row_block[S.Ai[p]];
                if (b < 0) continue;
                if (col_block[j] == -1) col_block[j] = b;
                else if (col_block[j] != b) return fail(S, TLPK_BADARG, "row_block is not block-angular: a column spans two blocks");
            }
    }
    S.nblocks = nblocks;

    pt.mark("graph of AA'");
    // ---- 3. adjacency graph of A*A' (original labels, no diagonal, both directions) ----
    std::vector<i64> xadj((size_t)m + 1, 0);
    std::vector<i32> adj;
    {
        // rows are independent: chunks of rows on the host threads, count then fill (the marker array of a
        // thread is stamped with 2k / 2k+1, so the two passes and the rows of a chunk never collide)
        constexpr i64 CH = 4096;
        const i64 nch = ((i64)m + CH - 1) / CH;
        const unsigned nthreads = host_threads(nch);
        std::vector<std::vector<i64>> t_mark(nthreads);
        auto sweep = [&](bool fill) {
            return parallel_for(nch, nthreads, [&](unsigned tid, i64 ch) {
                std::vector<i64> &mark = t_mark[tid];
                if (mark.empty()) mark.assign(m, -1);
                for (i32 k = (i32)(ch * CH); k < (i32)std::min<i64>(m, (ch + 1) * CH); ++k) {
                    const i64 stamp = 2 * (i64)k + (fill ? 1 : 0);
                    mark[k] = stamp;
                    i64 c = fill ? xadj[k] : 0;
                    for (i64 q = S.Tp[k]; q < S.Tp[k + 1]; ++q) {
                        const i32 j = S.Tj[q];
                        for (i64 p = S.Ap[j]; p < S.Ap[j + 1]; ++p) {
                            const i32 i = S.Ai[p];
                            if (mark[i] != stamp) { mark[i] = stamp; if (fill) adj[c] = i; ++c; }
                        }
                    }
                    if (!fill) xadj[k + 1] = c;
                }
            });
        };
        if (!sweep(false)) return fail(S, TLPK_OOM, "out of memory while building the graph of A*A'");
        for (i32 k = 0; k < m; ++k) xadj[k + 1] += xadj[k];
        adj.resize((size_t)xadj[m]);
        if (!sweep(true)) return fail(S, TLPK_OOM, "out of memory while building the graph of A*A'");
    }

    pt.mark("ordering (AMD)");
    // ---- 4. fill-reducing ordering ----
    std::vector<i32> order0;
    order0.reserve(m);
    std::vector<char> is_link(m, 0);
    if (opt.ordering == TLPK_ORDER_USER) {
        if (!opt.user_perm) return fail(S, TLPK_BADARG, "user_perm is null");
        std::vector<char> seen(m, 0);
        for (i32 i = 0; i < m; ++i) {
            const i64 v = opt.user_perm[i];
            if (v < 0 || v >= m || seen[v]) return fail(S, TLPK_BADARG, "user_perm is not a permutation");
            seen[v] = 1; order0.push_back((i32)v);
        }
        if (opt.row_block) return fail(S, TLPK_BADARG, "user_perm cannot be combined with row_block");
    } else if (!opt.row_block) {
        if (opt.ordering == TLPK_ORDER_NATURAL) { order0.resize(m); std::iota(order0.begin(), order0.end(), 0); }
        else amd_order(m, xadj, adj, order0);
    } else {
        // each diagonal block on its own (blocks are mutually non-adjacent in S); linking rows last
        std::vector<std::vector<i32>> members(nblocks);
        for (i32 i = 0; i < m; ++i) if (row_block[i] >= 0) members[row_block[i]].push_back(i);
        std::vector<i32> local(m, -1);
        for (i32 b = 0; b < nblocks; ++b) for (size_t t = 0; t < members[b].size(); ++t) local[members[b][t]] = (i32)t;
        // the blocks are independent graphs: ordered concurrently on the host's cores (the result does
        // not depend on the number of threads: every block is ordered by itself)
        std::vector<std::vector<i32>> border(nblocks);
        auto order_block = [&](i32 b) {
            const auto &mb = members[b];
            const i32 nb = (i32)mb.size();
            if (opt.ordering == TLPK_ORDER_NATURAL) { border[b].resize(nb); std::iota(border[b].begin(), border[b].end(), 0); return; }
            std::vector<i64> bx((size_t)nb + 1, 0);
            std::vector<i32> ba;
            for (i32 t = 0; t < nb; ++t) {
                const i32 v = mb[t];
                for (i64 p = xadj[v]; p < xadj[v + 1]; ++p) {
                    const i32 u = adj[p];
                    if (row_block[u] == b) ba.push_back(local[u]);
                }
                bx[t + 1] = (i64)ba.size();
            }
            amd_order(nb, bx, ba, border[b]);
        };
        {
            const unsigned nthreads = host_threads(nblocks);
            if (std::getenv("TLPK_TIMING")) std::fprintf(stderr, "[tlpk analyse] ordering %d blocks on %u threads\n", (int)nblocks, nthreads);
            if (!parallel_for(nblocks, nthreads, [&](unsigned, i64 b) { order_block((i32)b); }))
                return fail(S, TLPK_OOM, "out of memory while ordering the diagonal blocks");
        }
        for (i32 b = 0; b < nblocks; ++b) for (i32 t : border[b]) order0.push_back(members[b][t]);
        for (i32 i = 0; i < m; ++i) if (row_block[i] < 0) { order0.push_back(i); is_link[i] = 1; }
    }
    if ((i32)order0.size() != m) return fail(S, TLPK_INTERNAL, "ordering did not return a permutation");

    pt.mark("etree");
    // ---- 5. elimination tree, postorder ----
    std::vector<i32> iperm0(m);
    par_chunks(m, [&](i64 lo, i64 hi) { for (i64 i = lo; i < hi; ++i) iperm0[order0[i]] = (i32)i; });
    std::vector<i32> parent0;
    if (opt.row_block && nblocks >= 2 && opt.ordering != TLPK_ORDER_USER) {
        // block-angular: the ordering lists block after block, then the linking rows.  The nodes of a block
        // are adjacent only to their own block and to linking rows (which come later and are skipped by
        // the k < i test), so the blocks' ranges are independent; the linking rows follow sequentially.
        std::vector<i32> bstart;
        for (i32 i = 0; i < m; ++i)
            if (!is_link[order0[i]] && (i == 0 || row_block[order0[i]] != row_block[order0[i - 1]])) bstart.push_back(i);
        const i32 first_link0 = m - nlink;
        bstart.push_back(first_link0);
        parent0.assign(m, -1);
        std::vector<i32> anc(m, -1);
        const i64 nb = (i64)bstart.size() - 1;
        parallel_for_throw(nb, host_threads(nb), [&](unsigned, i64 b) { etree_range(bstart[b], bstart[b + 1], xadj, adj, order0, iperm0, parent0, anc); });
        etree_range(first_link0, m, xadj, adj, order0, iperm0, parent0, anc);
    } else
        etree_of(m, xadj, adj, order0, iperm0, parent0);
    // final order: postorder of the forest without the linking nodes, then the linking nodes
    std::vector<char> skip(m, 0);
    par_chunks(m, [&](i64 lo, i64 hi) { for (i64 i = lo; i < hi; ++i) skip[i] = is_link[order0[i]]; });
    std::vector<i32> post0;
    postorder_forest(m, parent0, nlink ? &skip : nullptr, post0);
    for (i32 i = 0; i < m; ++i) if (skip[i]) post0.push_back(i);
    if ((i32)post0.size() != m) return fail(S, TLPK_INTERNAL, "postorder lost nodes");
    S.perm.resize(m); S.iperm.resize(m);
    std::vector<i32> relabel(m);               // order0-label -> final label
    par_chunks(m, [&](i64 lo, i64 hi) { for (i64 k = lo; k < hi; ++k) { S.perm[k] = order0[post0[k]]; relabel[post0[k]] = (i32)k; } });
    par_chunks(m, [&](i64 lo, i64 hi) { for (i64 k = lo; k < hi; ++k) S.iperm[S.perm[k]] = (i32)k; });
    S.parent.assign(m, -1);
    par_chunks(m, [&](i64 lo, i64 hi) { for (i64 v = lo; v < hi; ++v) if (parent0[v] != -1) S.parent[relabel[v]] = relabel[parent0[v]]; });
    {
        std::atomic<int> bad_topo{0};
        par_chunks(m, [&](i64 lo, i64 hi) { for (i64 k = lo; k < hi; ++k) if (S.parent[k] != -1 && S.parent[k] <= k) bad_topo = 1; });
        if (bad_topo) return fail(S, TLPK_INTERNAL, "etree not topological");
    }
    const i32 first_link = m - nlink;

    pt.mark("pattern of S");
    // ---- 6. permuted lower pattern of S (rebuilt if the amalgamation re-orders columns) ----
    auto build_pattern = [&]() {
        // columns are independent: chunks of columns on the host threads
        constexpr i64 CH = 2048;
        const i64 nch = ((i64)m + CH - 1) / CH;
        const unsigned nthreads = host_threads(nch);
        S.Sp.assign((size_t)m + 1, 0);
        parallel_for_throw(nch, nthreads, [&](unsigned, i64 ch) {
            for (i32 kk = (i32)(ch * CH); kk < (i32)std::min<i64>(m, (ch + 1) * CH); ++kk) {
                const i32 k = S.perm[kk];
                i64 c = 1;
                for (i64 p = xadj[k]; p < xadj[k + 1]; ++p) if (S.iperm[adj[p]] > kk) ++c;
                S.Sp[kk + 1] = c;
            }
        });
        for (i32 kk = 0; kk < m; ++kk) S.Sp[kk + 1] += S.Sp[kk];
        S.nnzS = S.Sp[m];
        S.Si.resize((size_t)S.nnzS);
        parallel_for_throw(nch, nthreads, [&](unsigned, i64 ch) {
            for (i32 kk = (i32)(ch * CH); kk < (i32)std::min<i64>(m, (ch + 1) * CH); ++kk) {
                const i32 k = S.perm[kk];
                i64 q = S.Sp[kk];
                S.Si[q++] = kk;
                for (i64 p = xadj[k]; p < xadj[k + 1]; ++p) { const i32 ii = S.iperm[adj[p]]; if (ii > kk) S.Si[q++] = ii; }
                std::sort(S.Si.begin() + S.Sp[kk] + 1, S.Si.begin() + q);
            }
        });
    };
    build_pattern();

    pt.mark("column counts");
    // ---- 7. column counts (Gilbert, Ng & Peyton 1994: row-subtree leaves + LCA by union-find) ----
    {
        std::vector<i32> tpost;
        postorder_forest(m, S.parent, nullptr, tpost);        // a true postorder of the final tree
        std::vector<i32> pidx(m), first(m), delta(m, 0), maxfirst(m, -1), prevleaf(m, -1), anc(m);
        for (i32 k = 0; k < m; ++k) pidx[tpost[k]] = k;
        for (i32 v = 0; v < m; ++v) { first[v] = pidx[v]; anc[v] = v; }
        for (i32 k = 0; k < m; ++k) {                          // first descendant, children before parents
            const i32 v = tpost[k], p = S.parent[v];
            if (p != -1) first[p] = std::min(first[p], first[v]);
        }
        for (i32 k = 0; k < m; ++k) {
            const i32 j = tpost[k];
            delta[j] += (first[j] == k) ? 1 : 0;               // j is a leaf of the etree
            if (S.parent[j] != -1) delta[S.parent[j]]--;
            for (i64 p = S.Sp[j] + 1; p < S.Sp[j + 1]; ++p) {
                const i32 i = S.Si[p];                         // i is an ancestor of j, S[i,j] != 0
                if (first[j] <= maxfirst[i]) continue;         // j is not a leaf of row subtree T^i
                maxfirst[i] = first[j];
                const i32 jprev = prevleaf[i];
                prevleaf[i] = j;
                delta[j]++;
                if (jprev != -1) {
                    i32 q = jprev;
                    while (anc[q] != q) q = anc[q];
                    for (i32 s = jprev; s != q;) { const i32 t = anc[s]; anc[s] = q; s = t; }
                    delta[q]--;
                }
            }
            if (S.parent[j] != -1) anc[j] = S.parent[j];
        }
        S.colcount.assign(m, 0);
        for (i32 k = 0; k < m; ++k) {
            const i32 j = tpost[k];
            S.colcount[j] += delta[j];
            if (S.parent[j] != -1) S.colcount[S.parent[j]] += S.colcount[j];
        }
    }
    S.nnzL = 0; S.flops_chol = 0;
    for (i32 j = 0; j < m; ++j) {
        if (S.colcount[j] < 1) return fail(S, TLPK_INTERNAL, "column count < 1");
        S.nnzL += S.colcount[j]; S.flops_chol += (double)S.colcount[j] * (double)S.colcount[j];
    }

    pt.mark("fundamental supernodes");
    // ---- 8. fundamental supernodes ----
    // start[s] = first column.  j joins j-1 when parent[j-1] == j and the structures nest exactly
    // (count[j-1] == count[j] + 1).  Linking columns form one forced root front.
    std::vector<i32> sn_start;
    for (i32 j = 0; j < m; ++j) {
        bool join = false;
        if (j > 0) {
            if (j > first_link) join = true;                                 // inside the forced root
            else if (j == first_link) join = false;
            else join = (S.parent[j - 1] == j && S.colcount[j - 1] == S.colcount[j] + 1);
        }
        if (!join) sn_start.push_back(j);
    }
    sn_start.push_back(m);
    i32 ns_total = 0;
    std::vector<i32> &sparent = S.sparent_v;

    pt.mark("fronts");
    // ---- 9. supernodal tree and front row structures (for a given column partition) ----
    // rows == false: only the supernodal tree and the front sizes (exact for fundamental supernodes:
    // f = column count of the first column) -- all the amalgamation looks at; the row structures are
    // then built once, for the final partition
    auto build_fronts = [&](bool rows) -> int {
        ns_total = (i32)sn_start.size() - 1;
        S.nsuper = ns_total;
        S.sn_of_col.resize(m);
        par_chunks(ns_total, [&](i64 lo, i64 hi) { for (i64 s = lo; s < hi; ++s) for (i32 j = sn_start[s]; j < sn_start[s + 1]; ++j) S.sn_of_col[j] = (i32)s; });
        S.fronts.assign(ns_total, FrontDesc{});
        sparent.assign(ns_total, -1);
        for (i32 s = 0; s < ns_total; ++s) {
            // parent front: the one holding the etree parent of the supernode's last column
            const i32 pc = S.parent[sn_start[s + 1] - 1];
            sparent[s] = (pc == -1) ? -1 : S.sn_of_col[pc];
            if (sparent[s] != -1 && sparent[s] <= s) return fail(S, TLPK_INTERNAL, "supernodal tree not topological");
        }
        std::vector<i32> nchild(ns_total, 0);
        for (i32 s = 0; s < ns_total; ++s) if (sparent[s] != -1) nchild[sparent[s]]++;
        i32 acc = 0;
        for (i32 s = 0; s < ns_total; ++s) { S.fronts[s].child_ptr = acc; S.fronts[s].nchild = 0; acc += nchild[s]; }
        S.children.assign(acc, -1);
        for (i32 s = 0; s < ns_total; ++s) if (sparent[s] != -1) {
            FrontDesc &p = S.fronts[sparent[s]];
            S.children[p.child_ptr + p.nchild++] = s;
        }
        // rows(s) = cols(s) ++ sorted union of the below-diagonal structure of every column of s
        // and of the children's rows; merged (relaxed) supernodes carry explicit zeros.
        if (!rows) {
            for (i32 s = 0; s < ns_total; ++s) {
                FrontDesc &w = S.fronts[s];
                const i32 j0 = sn_start[s];
                w.ns = sn_start[s + 1] - j0; w.f = (i32)S.colcount[j0]; w.col0 = j0; w.parent = sparent[s];
            }
            return TLPK_OK;
        }
        // A front needs the below-rows of its children: level by level, deepest first, the fronts of a
        // level on the host threads (each with its own marker array), then one sequential pass lays the
        // lists out in front order.
        std::vector<std::vector<i32>> below(ns_total);
        {
            std::vector<i32> sdepth(ns_total, 0);
            i32 maxd = 0;
            for (i32 s = ns_total - 1; s >= 0; --s) { sdepth[s] = (sparent[s] == -1) ? 0 : sdepth[sparent[s]] + 1; maxd = std::max(maxd, sdepth[s]); }
            std::vector<std::vector<i32>> bylevel(maxd + 1);
            for (i32 s = 0; s < ns_total; ++s) bylevel[sdepth[s]].push_back(s);
            const unsigned nthreads = host_threads(ns_total);
            std::vector<std::vector<i32>> t_mark(nthreads);
            for (i32 d = maxd; d >= 0; --d) {
                const std::vector<i32> &lv = bylevel[d];
                const bool ok = parallel_for((i64)lv.size(), std::min<unsigned>(nthreads, (unsigned)std::max<size_t>(1, lv.size())), [&](unsigned tid, i64 q) {
                    const i32 s = lv[q];
                    std::vector<i32> &mark = t_mark[tid];
                    if (mark.empty()) mark.assign(m, -1);
                    const i32 j0 = sn_start[s], j1 = sn_start[s + 1] - 1;
                    std::vector<i32> &tmp = below[s];
                    for (i32 j = j0; j <= j1; ++j)
                        for (i64 p = S.Sp[j] + 1; p < S.Sp[j + 1]; ++p) {
                            const i32 i = S.Si[p];
                            if (i > j1 && mark[i] != s) { mark[i] = s; tmp.push_back(i); }
                        }
                    const FrontDesc &fd = S.fronts[s];
                    for (i32 t = 0; t < fd.nchild; ++t)
                        for (const i32 i : below[S.children[fd.child_ptr + t]])
                            if (i > j1 && mark[i] != s) { mark[i] = s; tmp.push_back(i); }
                    std::sort(tmp.begin(), tmp.end());
                }, 16);
                if (!ok) return fail(S, TLPK_OOM, "out of memory while building the front structures");
            }
        }
        S.rowidx.clear();
        {
            size_t total = (size_t)m;
            for (i32 s = 0; s < ns_total; ++s) total += below[s].size();
            S.rowidx.reserve(total);
        }
        S.max_front = 0;
        for (i32 s = 0; s < ns_total; ++s) {
            const i32 j0 = sn_start[s], j1 = sn_start[s + 1] - 1;
            const std::vector<i32> &tmp = below[s];
            FrontDesc &w = S.fronts[s];
            w.rowoff = (i64)S.rowidx.size();
            w.ns = j1 - j0 + 1;
            w.f = w.ns + (i32)tmp.size();
            w.col0 = j0;
            w.parent = sparent[s];
            for (i32 j = j0; j <= j1; ++j) S.rowidx.push_back(j);
            S.rowidx.insert(S.rowidx.end(), tmp.begin(), tmp.end());
            if (w.f < S.colcount[j0]) return fail(S, TLPK_INTERNAL, "front smaller than its first column count");
            if (!tmp.empty() && sparent[s] == -1) return fail(S, TLPK_INTERNAL, "root front with rows below");
            if (!tmp.empty() && S.sn_of_col[tmp[0]] != sparent[s]) return fail(S, TLPK_INTERNAL, "first below-row is not in the parent front");
            S.max_front = std::max<i64>(S.max_front, w.f);
        }
        return TLPK_OK;
    };
    const bool will_relax = opt.relax && (i32)sn_start.size() - 1 > 1;
    { const int rc = build_fronts(!will_relax); if (rc != TLPK_OK) return rc; }

    pt.mark("amalgamation");
    // ---- 9b. relaxed amalgamation (any child, not only the adjacent one) ----
    // A child front c is merged into its parent p when either the explicit zeros stay small
    // (CHOLMOD-style width classes) or -- the multifrontal criterion -- padding c's ns_c columns
    // to the parent's rows costs fewer flops than shipping its rs_c x rs_c update matrix through
    // HBM would cost in time (extra_flops < GAMMA * rs_c^2, GAMMA ~ flop rate x bytes per entry /
    // bandwidth) without growing memory.  Merged members become contiguous by re-ordering the
    // columns with another topological order of the same elimination tree (identical fill).
    if (opt.relax && ns_total > 1) {
        double GAMMA = 25.0;
        if (const char *e = std::getenv("TLPK_RELAX_GAMMA")) GAMMA = std::atof(e);    // tuning knob
        double GAMMA_TALL = 400.0, TALL_RATIO = 0.5;                                  // (read once: two getenv calls per candidate were a fifth of this phase)
        if (const char *e = std::getenv("TLPK_RELAX_GAMMA_TALL")) GAMMA_TALL = std::atof(e);
        if (const char *e = std::getenv("TLPK_RELAX_TALL_RATIO")) TALL_RATIO = std::atof(e);
        const i32 forced_root = nlink ? ns_total - 1 : -1;
        std::vector<i32> into(ns_total, -1);
        std::vector<double> cns(ns_total), cf(ns_total), cz(ns_total, 0.0);
        // children lists as flat arrays (400 000 fronts on the north-star LP: a vector per front was a third of this phase): the initial lists in
        // `kid0` (children in ascending order), the list a processed front KEEPS appended to `kept`; kid_list(x) = whichever is current
        std::vector<i32> kid0_ptr((size_t)ns_total + 1, 0), kid0((size_t)ns_total), kept, kept_ptr((size_t)ns_total, -1), kept_cnt((size_t)ns_total, 0);
        for (i32 s = 0; s < ns_total; ++s) {
            cns[s] = S.fronts[s].ns; cf[s] = S.fronts[s].f;
            if (sparent[s] != -1) kid0_ptr[(size_t)sparent[s] + 1]++;
        }
        for (i32 s = 0; s < ns_total; ++s) kid0_ptr[(size_t)s + 1] += kid0_ptr[(size_t)s];
        {
            std::vector<i32> fill(kid0_ptr.begin(), kid0_ptr.end() - 1);
            for (i32 s = 0; s < ns_total; ++s) if (sparent[s] != -1) kid0[(size_t)fill[(size_t)sparent[s]]++] = s;
        }
        kept.reserve((size_t)ns_total);
        auto kid_list = [&](i32 x, const i32 *&first, i32 &count) {
            if (kept_ptr[(size_t)x] >= 0) { first = kept.data() + kept_ptr[(size_t)x]; count = kept_cnt[(size_t)x]; }
            else { first = kid0.data() + kid0_ptr[(size_t)x]; count = kid0_ptr[(size_t)x + 1] - kid0_ptr[(size_t)x]; }
        };
        std::vector<i32> cand, keep;
        bool any_merge = false;
        for (i32 p = 0; p < ns_total; ++p) {
            if (p == forced_root || kid0_ptr[(size_t)p + 1] == kid0_ptr[(size_t)p]) continue;
            cand.assign(kid0.begin() + kid0_ptr[(size_t)p], kid0.begin() + kid0_ptr[(size_t)p + 1]); keep.clear();
            std::sort(cand.begin(), cand.end(), [&](i32 a, i32 b) { return (cf[a] - cns[a]) > (cf[b] - cns[b]); });
            for (size_t idx = 0; idx < cand.size(); ++idx) {
                const i32 c = cand[idx];
                const double nc = cns[c], fc = cf[c], np = cns[p], fp = cf[p];
                const double rsc = fc - nc, fnew = nc + fp;
                const double extra_zeros = nc * (fnew - fc);
                const double newz = cz[c] + cz[p] + extra_zeros;
                const double total = (nc + np) * fnew;
                const double width = nc + np;
                const double extra_flops = nc * (fnew * fnew - fc * fc);
                // the zero-fraction classes bound MEMORY; the last, unbounded-width class also needs a bound on the padded FLOPS
                // (a 5-row leaf column absorbed by a 4 500-row front costs 2e7 flops of zeros for 25 flops of work)
                static const double ZFRAC = [] { const char *e = std::getenv("TLPK_RELAX_ZFRAC"); return e ? std::atof(e) : 0.05; }();
                static const double AFLOPS = [] { const char *e = std::getenv("TLPK_RELAX_AFLOPS"); return e ? std::atof(e) : 1e300; }();
                static const double AGAMMA = [] { const char *e = std::getenv("TLPK_RELAX_AGAMMA"); return e ? std::atof(e) : 0.0; }();
                const bool rule_a = extra_zeros == 0 || width <= 4 || (width <= 16 && newz <= 0.8 * total) ||
                                    (width <= 48 && newz <= 0.1 * total) ||
                                    (newz <= ZFRAC * total && extra_flops <= AFLOPS + AGAMMA * rsc * rsc);
                const bool rule_b = extra_flops < GAMMA * rsc * rsc && extra_zeros < 0.5 * rsc * rsc;
                // a child almost as tall as its parent (a thin front with thousands of rows) ships an
                // update matrix of ~fp^2 entries for a handful of columns: merging pads little
                // (measured: C4 63.7 -> 62.2 ms/step, headline instance 182 -> 172 ms/step, extend-add 30 -> 11 ms;
                // saturates above ~400)
                const bool rule_c = rsc >= TALL_RATIO * fp && extra_flops < GAMMA_TALL * rsc * rsc && extra_zeros < 0.5 * rsc * rsc;
                if (rule_a || rule_b || rule_c) {
                    into[c] = p; cns[p] += nc; cf[p] = fnew; cz[p] = newz; any_merge = true;
                    const i32 *gf; i32 gc;
                    kid_list(c, gf, gc);
                    cand.insert(cand.end(), gf, gf + gc);          // grandchildren now hang off p
                } else {
                    keep.push_back(c);
                }
            }
            kept_ptr[(size_t)p] = (i32)kept.size(); kept_cnt[(size_t)p] = (i32)keep.size();
            kept.insert(kept.end(), keep.begin(), keep.end());
        }
        pt.mark("amalgamation: re-order");
        if (any_merge) {
            auto root_of = [&](i32 s) { while (into[s] != -1) s = into[s]; return s; };
            // members of every group (ascending), flat
            std::vector<i32> grp((size_t)ns_total), mem_ptr((size_t)ns_total + 1, 0), mem((size_t)ns_total);
            for (i32 s = 0; s < ns_total; ++s) { grp[(size_t)s] = root_of(s); mem_ptr[(size_t)grp[(size_t)s] + 1]++; }
            for (i32 s = 0; s < ns_total; ++s) mem_ptr[(size_t)s + 1] += mem_ptr[(size_t)s];
            {
                std::vector<i32> fill(mem_ptr.begin(), mem_ptr.end() - 1);
                for (i32 s = 0; s < ns_total; ++s) mem[(size_t)fill[(size_t)grp[(size_t)s]]++] = s;
            }
            // post-order over the group tree (children groups before the group's own columns)
            std::vector<i32> newpos(m, -1), new_start;
            i32 counter = 0;
            std::vector<std::pair<i32, size_t>> stack;
            for (i32 r = 0; r < ns_total; ++r) {
                if (into[r] != -1 || sparent[r] != -1) continue;     // group roots without a parent
                stack.emplace_back(r, 0);
                while (!stack.empty()) {
                    auto &top = stack.back();
                    const i32 g = top.first;
                    const i32 *kf; i32 kc;
                    kid_list(g, kf, kc);
                    if (top.second < (size_t)kc) { const i32 ch = kf[top.second++]; stack.emplace_back(ch, 0); continue; }
                    new_start.push_back(counter);
                    for (i32 q = mem_ptr[(size_t)g]; q < mem_ptr[(size_t)g + 1]; ++q) {
                        const i32 mm = mem[(size_t)q];
                        for (i32 j = sn_start[mm]; j < sn_start[mm + 1]; ++j) newpos[j] = counter++;
                    }
                    stack.pop_back();
                }
            }
            if (counter != m) return fail(S, TLPK_INTERNAL, "amalgamation re-ordering lost columns");
            new_start.push_back(m);
            std::vector<i32> perm2(m), parent2(m, -1), cc2(m);
            par_chunks(m, [&](i64 lo, i64 hi) {
                for (i64 k = lo; k < hi; ++k) {
                    perm2[newpos[k]] = S.perm[k];
                    cc2[newpos[k]] = S.colcount[k];
                    if (S.parent[k] != -1) parent2[newpos[k]] = newpos[S.parent[k]];
                }
            });
            S.perm.swap(perm2); S.parent.swap(parent2); S.colcount.swap(cc2);
            par_chunks(m, [&](i64 lo, i64 hi) { for (i64 k = lo; k < hi; ++k) S.iperm[S.perm[k]] = (i32)k; });
            {
                std::atomic<int> bad_topo{0};
                par_chunks(m, [&](i64 lo, i64 hi) { for (i64 k = lo; k < hi; ++k) if (S.parent[k] != -1 && S.parent[k] <= k) bad_topo = 1; });
                if (bad_topo) return fail(S, TLPK_INTERNAL, "re-ordered etree not topological");
            }
            if (nlink) for (i32 k = first_link; k < m; ++k) if (!is_link[S.perm[k]]) return fail(S, TLPK_INTERNAL, "linking rows moved");
            sn_start.swap(new_start);
            pt.mark("amalgamation: pattern");
            build_pattern();
        }
        pt.mark("amalgamation: fronts");
        const int rc = build_fronts(true);
        if (rc != TLPK_OK) return rc;
    }
    { std::vector<i32>().swap(adj); std::vector<i64>().swap(xadj); }
    S.nlink_v = nlink;
    pt.mark(nullptr);
    return TLPK_OK;
}

int analyse_rank(Symbolic &S, const Options &opt) {
    PhaseTimer pt;
    AnalysePoolScope pool_scope;
    g_host_thread_div = opt.analyse_div > 0 ? opt.analyse_div : std::max<i32>(1, opt.nranks);
    const i32 m = (i32)S.m, n = (i32)S.n;
    const std::vector<i32> &row_block = S.row_block_v, &col_block = S.col_block_v, &sparent = S.sparent_v;
    const i32 nblocks = S.nblocks, nlink = S.nlink_v, ns_total = S.nsuper;
    if (opt.nranks < 1 || opt.rank < 0 || opt.rank >= opt.nranks) return fail(S, TLPK_BADARG, "bad rank/nranks");
    if (opt.nranks > 1 && row_block.empty()) return fail(S, TLPK_BADARG, "sharding needs row_block (general sparse LPs are single-GPU)");
    const bool have_blocks = !row_block.empty();

    pt.mark("levels");
    // ---- 10. depths, levels ----
    S.depth.assign(ns_total, 0);
    for (i32 s = ns_total - 1; s >= 0; --s) S.depth[s] = (sparent[s] == -1) ? 0 : S.depth[sparent[s]] + 1;
    S.nlevels = 0;
    for (i32 s = 0; s < ns_total; ++s) S.nlevels = std::max(S.nlevels, S.depth[s] + 1);
    S.level_ptr.assign((size_t)S.nlevels + 1, 0);
    for (i32 s = 0; s < ns_total; ++s) S.level_ptr[S.depth[s] + 1]++;
    for (i32 d = 0; d < S.nlevels; ++d) S.level_ptr[d + 1] += S.level_ptr[d];
    S.level_fronts.resize(ns_total);
    {
        std::vector<i32> cur(S.level_ptr.begin(), S.level_ptr.end() - 1);
        for (i32 s = 0; s < ns_total; ++s) S.level_fronts[cur[S.depth[s]]++] = s;
    }

    pt.mark("ownership");
    // ---- 11. ownership (block-angular sharding) ----
    S.front_block.assign(ns_total, -1);
    S.front_local.assign(ns_total, 1);
    S.root_front = (nlink > 0) ? ns_total - 1 : -1;
    std::vector<i32> block_owner(std::max(nblocks, 1), 0);
    if (have_blocks) {
        std::vector<double> bflops(nblocks, 0.0);
        for (i32 s = 0; s < ns_total; ++s) {
            const i32 b = row_block[S.perm[S.fronts[s].col0]];
            S.front_block[s] = b;
            if (b >= 0) {
                const double f = S.fronts[s].f, k = S.fronts[s].ns;
                bflops[b] += k * f * f;     // proportional weight
                if (S.fronts[s].parent != -1 && S.front_block[s] < 0) return fail(S, TLPK_INTERNAL, "linking front below the root");
            } else if (s != S.root_front) return fail(S, TLPK_INTERNAL, "linking column outside the root front");
        }
        // contiguous block ranges balanced by weight: block b goes to the rank whose share of
        // the cumulative weight contains b's midpoint; never more ranks than blocks in use
        double total = 0; for (double f : bflops) total += f;
        double acc = 0;
        for (i32 b = 0; b < nblocks; ++b) {
            i32 r = (total > 0) ? (i32)((acc + 0.5 * bflops[b]) / total * opt.nranks) : (i32)((i64)b * opt.nranks / nblocks);
            r = std::max(0, std::min(r, opt.nranks - 1));
            if (b > 0) r = std::max(r, block_owner[b - 1]);
            block_owner[b] = r;
            acc += bflops[b];
        }
        S.n_local_blocks = 0;
        for (i32 b = 0; b < nblocks; ++b) if (block_owner[b] == opt.rank) S.n_local_blocks++;
        for (i32 s = 0; s < ns_total; ++s) {
            const i32 b = S.front_block[s];
            S.front_local[s] = (b < 0) || (block_owner[b] == opt.rank);
        }
    }

    // stream groups: the diagonal blocks are independent subtrees below the root front; block b runs
    // on stream b % ngroups so that one group's latency-bound steps (potrf/trsm chains, diagonal
    // solves) overlap the other groups' MFMA updates.  General sparse LPs: one group.
    S.ngroups = 1;
    if (have_blocks && nblocks >= 2) {
        // 2 groups x (stream + side stream) = 4 streams = the runtime's default number of hardware
        // queues; more streams share queues and serialise (measured: 2 -> 69.5, 3 -> 75.5, 4 -> 74.3 ms/step on C4)
        S.ngroups = std::min(2, nblocks);
        // a rank of a sharded / multi-device handle that owns few blocks: every launch of a group holds one tile set per block, two groups halve it for
        // nothing to overlap with (round 5, rank-local step of C4 / north-star shape: 8 blocks 12.5 vs 13.0 ms, 12 blocks 25.7 vs 25.7, 16 blocks 18.8 vs
        // 18.9, 25 blocks 42.6 vs 42.0, 32 blocks 30.4 vs 30.1 with one / two groups; results do not depend on the number of groups)
        if (opt.nranks > 1 && S.n_local_blocks <= 8) S.ngroups = 1;
        if (opt.streams > 0) S.ngroups = std::min({opt.streams, MAX_GROUPS, nblocks});
        else if (const char *e = std::getenv("TLPK_STREAMS")) S.ngroups = std::max(1, std::min({std::atoi(e), MAX_GROUPS, nblocks}));
    }
    S.front_group.assign(ns_total, 0);
    for (i32 s = 0; s < ns_total; ++s) if (S.front_block[s] >= 0) S.front_group[s] = S.front_block[s] % S.ngroups;

    // a rank only ever walks its own children: drop the other ranks' block roots from the
    // (replicated) root front's child list
    if (opt.nranks > 1)
        for (i32 s = 0; s < ns_total; ++s) {
            FrontDesc &w = S.fronts[s];
            i32 kept = 0;
            for (i32 t = 0; t < w.nchild; ++t) {
                const i32 c = S.children[w.child_ptr + t];
                if (S.front_local[c]) S.children[w.child_ptr + kept++] = c;
            }
            w.nchild = kept;
        }
    S.col_local.assign(n, 1);
    S.row_local.assign(m, 1);
    if (have_blocks && opt.nranks > 1) {
        for (i32 j = 0; j < n; ++j) {
            const i32 b = col_block[j];
            S.col_local[j] = (b < 0) ? (opt.rank == 0) : (block_owner[b] == opt.rank);
        }
        for (i32 i = 0; i < m; ++i) {
            const i32 b = row_block[i];
            S.row_local[i] = (b < 0) ? 2 : (block_owner[b] == opt.rank);   // 2 = linking (replicated)
        }
    } else if (have_blocks) {
        for (i32 i = 0; i < m; ++i) if (row_block[i] < 0) S.row_local[i] = 2;
    }

    pt.mark("offsets");
    // ---- 12. storage offsets (local fronts only) ----
    S.lval_len = 0; S.uc_len = 0; S.ubuf_len[0] = S.ubuf_len[1] = 0; S.dinv_len = 0;
    for (i32 s = 0; s < ns_total; ++s) {
        FrontDesc &w = S.fronts[s];
        w.ubuf = S.depth[s] & 1;
        if (!S.front_local[s]) { w.lda = w.f; w.loff = -1; w.uoff = -1; w.ucoff = -1; w.dinvoff = -1; continue; }
        // Panel columns of the larger fronts start on 128-byte lines: the kernels stream 64..256 contiguous rows of a
        // column per load, and a misaligned 512-byte segment touches 5 lines instead of 4 (measured: 28 % more
        // fabric traffic in the solve sweeps than algorithmic bytes with ld = f).
        w.lda = (w.f >= LDA_PAD_MIN_F) ? (w.f + 15) / 16 * 16 : w.f;
        if (w.lda != w.f || w.f >= LDA_PAD_MIN_F) S.lval_len = (S.lval_len + 15) / 16 * 16;
        w.loff = S.lval_len; S.lval_len += pk_len(w.lda, w.ns);
        w.ucoff = S.uc_len; S.uc_len += (w.f - w.ns);
        if (w.ns >= NB_IN) S.dinv_len = (S.dinv_len + 15) / 16 * 16;     // the inverted 64 x 64 blocks start on 128-byte lines (trsm_task_dma loads them 16 bytes per lane)
        w.dinvoff = S.dinv_len;
        S.dinv_len += (w.ns >= NB_IN) ? (i64)((w.ns + NB_IN - 1) / NB_IN) * NB_IN * NB_IN : (i64)w.ns * w.ns;
    }
    {
        // update-matrix buffers: ping-pong by depth parity inside each stream group (groups are not
        // level-synchronised with each other, so every group gets its own region of both buffers)
        std::vector<std::array<i64, 2>> gmax(S.ngroups, {0, 0});
        std::vector<i64> off;
        for (i32 d = 0; d < S.nlevels; ++d) {
            off.assign(S.ngroups, 0);
            for (i32 t = S.level_ptr[d]; t < S.level_ptr[d + 1]; ++t) {
                const i32 s = S.level_fronts[t];
                FrontDesc &w = S.fronts[s];
                if (!S.front_local[s]) continue;
                const i64 rs = w.f - w.ns;
                const i32 g = S.front_group[s];
                w.uoff = off[g]; off[g] += rs * rs;           // group-relative for now
            }
            for (i32 g = 0; g < S.ngroups; ++g) gmax[g][d & 1] = std::max(gmax[g][d & 1], off[g]);
        }
        std::array<i64, 2> base = {0, 0};
        std::vector<std::array<i64, 2>> gbase(S.ngroups);
        for (i32 g = 0; g < S.ngroups; ++g) { gbase[g] = base; base[0] += gmax[g][0]; base[1] += gmax[g][1]; }
        S.ubuf_len[0] = base[0]; S.ubuf_len[1] = base[1];
        for (i32 s = 0; s < ns_total; ++s) {
            FrontDesc &w = S.fronts[s];
            if (S.front_local[s]) w.uoff += gbase[S.front_group[s]][S.depth[s] & 1];
        }
    }
    S.flops_panel = 0;
    for (i32 s = 0; s < ns_total; ++s) {
        const double f = S.fronts[s].f, k = S.fronts[s].ns;
        // potrf k^3/3 + trsm (f-k)k^2 + syrk (f-k)^2 k, in flops (multiply-add = 2)
        S.flops_panel += k * k * k / 3.0 + (f - k) * k * k + (f - k) * (f - k) * k;
    }

    // Algorithmic flops of the left-looking MFMA update (k_update), in the CHOLMOD `fl` convention that
    // defines flops_chol = sum_j l_j^2 (l_j = true nnz of column j of L, no amalgamation zeros): column
    // j's l_j^2 flops update the l_j trailing rows; the part whose TARGET column lies in the same
    // NB_OUT-wide block column of the front is done by the potrf / trsm kernels, the rest -- targets in
    // later block columns and in the update matrix, (l_j - r_j)^2 with r_j = columns left in j's block
    // column, itself included -- by k_update.  This is the numerator of bench.py's roofline.frac.
    S.flops_update_alg = 0;
    for (i32 s = 0; s < ns_total; ++s) {
        const FrontDesc &w = S.fronts[s];
        if (!S.front_local[s]) continue;
        for (i32 c = 0; c < w.ns; ++c) {
            const i32 r = std::min((c / NB_OUT + 1) * NB_OUT, w.ns) - c;
            const double l = (double)S.colcount[w.col0 + c] - (double)r;
            if (l > 0) S.flops_update_alg += l * l;
        }
    }

    pt.mark("relative indices");
    // ---- 13. relative indices (below-rows of each front -> position in the parent front) ----
    {
        // offsets first (prefix sum), then every front searches its parent on the host threads
        i64 acc = 0;
        for (i32 s = 0; s < ns_total; ++s) { FrontDesc &w = S.fronts[s]; w.reloff = acc; if (w.parent != -1) acc += w.f - w.ns; }
        S.rel.assign((size_t)acc, 0);
        std::atomic<int> bad{0};
        parallel_for_throw(ns_total, host_threads(ns_total), [&](unsigned, i64 s) {
            const FrontDesc &w = S.fronts[s];
            if (w.parent == -1) return;
            const FrontDesc &p = S.fronts[w.parent];
            i64 qp = p.rowoff, out = w.reloff;
            const i64 qend = p.rowoff + p.f;
            for (i64 q = w.rowoff + w.ns; q < w.rowoff + w.f; ++q) {
                const i32 r = S.rowidx[q];
                while (qp < qend && S.rowidx[qp] < r) ++qp;
                if (qp == qend || S.rowidx[qp] != r) { bad = 1; return; }
                S.rel[out++] = (i32)(qp - p.rowoff);
            }
        }, 64);
        if (bad) return fail(S, TLPK_INTERNAL, "child row missing from parent front");
    }
    // ---- 13a. extend-add lookup: the parent's columns are cut into the ranges of its extend-add workgroups
    // (ea_cols wide from 0 inside the pivot columns, and from ns inside the update-matrix columns); for every
    // range boundary the child stores the first of its columns that lands at or after it, so that a workgroup
    // finds "the child's columns in my range" with two loads instead of two binary searches in HBM.
    {
        // fronts whose panel is formed by k_front_assemble (see fa_min_f above)
        S.front_fa.assign((size_t)ns_total, 0);
        for (i32 s = 0; s < ns_total; ++s) {
            const FrontDesc &w = S.fronts[s];
            if (!S.front_local[s] || w.f < fa_min_f() || w.nchild == 0 || (w.f == 1 && w.ns == 1)) continue;
            double contrib = 0;
            for (i32 t = 0; t < w.nchild; ++t) { const FrontDesc &cd = S.fronts[S.children[w.child_ptr + t]]; const double r = cd.f - cd.ns; contrib += 0.5 * r * r; }
            if (contrib >= fa_density() * 0.5 * (double)w.f * w.f && w.ns >= FA_CW) S.front_fa[(size_t)s] = 1;     // contributions per entry of the front
        }
        i64 acc = 0;
        for (i32 s = 0; s < ns_total; ++s) {
            FrontDesc &w = S.fronts[s];
            w.eatab = -1;
            if (w.parent == -1) continue;
            const FrontDesc &p = S.fronts[w.parent];
            const bool pfa = S.front_fa[(size_t)w.parent];
            if (acc > (i64)INT32_MAX - (ea_nbounds(p, pfa) + 1)) return fail(S, TLPK_TOO_LARGE, "extend-add lookup table exceeds 2^31 entries");
            w.eatab = (i32)acc;
            acc += ea_nbounds(p, pfa);
        }
        S.ea_tab.assign((size_t)acc, 0);
        parallel_for_throw(ns_total, host_threads(ns_total), [&](unsigned, i64 s) {
            const FrontDesc &w = S.fronts[s];
            if (w.parent == -1) return;
            const FrontDesc &p = S.fronts[w.parent];
            const bool pfa = S.front_fa[(size_t)w.parent];
            const i32 rsc = w.f - w.ns, nb = ea_nbounds(p, pfa);
            const i32 *rel = S.rel.data() + w.reloff;
            i32 q = 0;
            for (i32 k = 0; k < nb; ++k) {
                const i32 bound = ea_bound(p, pfa, k);
                while (q < rsc && rel[q] < bound) ++q;
                S.ea_tab[(size_t)w.eatab + k] = q;
            }
        }, 64);
    }

    pt.mark("structural zeros");
    // ---- 13c. structural zeros of the amalgamated fronts.  A merged supernode is stored and factorised as a dense trapezoid, but
    // the columns of an absorbed child are zero outside the child's own row structure: on the north-star instance a third of the
    // flops of the left-looking update multiplied such zeros.  For every front with padding: one bit per (16-column K slab, 16-row
    // group) = "some column of the slab has a TRUE nonzero of L in these rows"; build_schedule gives every update tile the list of
    // K slabs in which BOTH of its operand row ranges have one.  True structures come from the column elimination tree: inside a front the
    // columns form chains with nested structures (parent[j-1] == j and count[j-1] == count[j] + 1: struct(j) = struct(j-1) \ {j-1}), the
    // structure of a chain head is its column of S, the structures of the etree children of the chain's columns that lie in the front,
    // and the rows below the child FRONTS that enter the front at a column of the chain (the last column of a front holds all rows
    // below the front).  Bits of rows above a column are never consulted (operand rows lie below the K columns), so plain ORs do.
    {
        i64 min_f = 256;
        bool on = true;
        if (const char *e = std::getenv("TLPK_SKIP")) on = std::atoi(e) != 0;                  // TLPK_SKIP=0: no skip lists
        if (const char *e = std::getenv("TLPK_SKIP_MIN_F")) min_f = std::atoll(e);             // testing knob
        S.skip_off.assign((size_t)ns_total, -1);
        S.skip_bits.clear();
        std::vector<i32> elig;
        i64 acc = 0;
        if (on)
            for (i32 s = 0; s < ns_total; ++s) {
                const FrontDesc &w = S.fronts[s];
                if (!S.front_local[s] || w.ns < 2 * 16 || w.f < min_f || w.f <= w.ns) continue;
                i64 stored = 0, truth = 0;
                for (i32 c = 0; c < w.ns; ++c) { stored += w.f - c; truth += S.colcount[w.col0 + c]; }
                if (stored == truth) continue;                 // no padding: nothing to skip
                const i64 nsl = (w.ns + 15) / 16, W = ((w.f + 15) / 16 + 63) / 64;
                S.skip_off[s] = acc; acc += nsl * W;
                elig.push_back(s);
            }
        S.skip_bits.assign((size_t)acc, 0);
        std::atomic<int> bad{0};
        parallel_for_throw((i64)elig.size(), host_threads((i64)elig.size()), [&](unsigned, i64 q) {
            const i32 s = elig[(size_t)q];
            const FrontDesc &w = S.fronts[s];
            const i32 ns = w.ns, f = w.f, col0 = w.col0;
            const i64 W = ((f + 15) / 16 + 63) / 64;
            std::vector<i32> head(ns), hid(ns, -1);
            i32 nheads = 0;
            for (i32 c = 0; c < ns; ++c) {
                const bool chain = c > 0 && S.parent[col0 + c - 1] == col0 + c && S.colcount[col0 + c - 1] == S.colcount[col0 + c] + 1;
                head[c] = chain ? head[c - 1] : c;
                if (!chain) hid[c] = nheads++;
            }
            std::vector<uint64_t> B((size_t)nheads * W, 0);
            auto setpos = [&](uint64_t *b, i32 pos) { b[(pos >> 4) >> 6] |= (uint64_t)1 << ((pos >> 4) & 63); };
            const i32 *below = S.rowidx.data() + w.rowoff + ns;
            for (i32 t = 0; t < w.nchild; ++t) {
                const FrontDesc &cd = S.fronts[S.children[w.child_ptr + t]];
                const i32 p = S.parent[cd.col0 + cd.ns - 1];
                if (p < col0 || p >= col0 + ns) { bad = 1; return; }
                uint64_t *hb = B.data() + (size_t)hid[head[p - col0]] * W;
                for (i32 r = 0; r < cd.f - cd.ns; ++r) setpos(hb, S.rel[cd.reloff + r]);
            }
            for (i32 c = 0; c < ns; ++c) {
                const i32 j = col0 + c;
                uint64_t *hb = B.data() + (size_t)hid[head[c]] * W;
                if (head[c] == c)
                    for (i64 e = S.Sp[j]; e < S.Sp[j + 1]; ++e) {
                        const i32 i = S.Si[e];
                        i32 pos;
                        if (i < col0 + ns) pos = i - col0;
                        else {
                            const i32 *it = std::lower_bound(below, below + (f - ns), i);
                            if (it == below + (f - ns) || *it != i) { bad = 1; return; }
                            pos = ns + (i32)(it - below);
                        }
                        setpos(hb, pos);
                    }
                const i32 p = S.parent[j];
                if (p != -1 && p < col0 + ns && head[p - col0] != head[c]) {
                    uint64_t *pb = B.data() + (size_t)hid[head[p - col0]] * W;
                    for (i64 x = 0; x < W; ++x) pb[x] |= hb[x];
                }
            }
            uint64_t *out = S.skip_bits.data() + S.skip_off[s];
            for (i32 c = 0; c < ns; ++c) {
                if (c > 0 && head[c] == head[c - 1] && (c & 15) != 0) continue;      // same chain as the previous column of this slab
                const uint64_t *hb = B.data() + (size_t)hid[head[c]] * W;
                uint64_t *ob = out + (size_t)(c >> 4) * W;
                for (i64 x = 0; x < W; ++x) ob[x] |= hb[x];
            }
        }, 1);
        if (bad) return fail(S, TLPK_INTERNAL, "structural-zero analysis: inconsistent front structure");
    }

    pt.mark("gather lists");
    // ---- 13b. forward-solve gather lists: for every row t of a front, the entries of its
    // children's contribution vectors that land on it, in child order (the order the sums are
    // taken in).  One thread per row then gathers without conflicts.
    {
        const i64 nrow_total = (i64)S.rowidx.size();
        S.gth_ptr.assign(nrow_total + 1, 0);
        const unsigned nthreads = host_threads(ns_total);
        // the rows [rowoff, rowoff + f) of a front belong to that front alone: fronts on the host threads
        parallel_for_throw(ns_total, nthreads, [&](unsigned, i64 s) {
            const FrontDesc &w = S.fronts[s];
            if (!S.front_local[s]) return;
            for (i32 t = 0; t < w.nchild; ++t) {
                const FrontDesc &cd = S.fronts[S.children[w.child_ptr + t]];
                const i32 rsc = cd.f - cd.ns;
                for (i32 r = 0; r < rsc; ++r) ++S.gth_ptr[w.rowoff + S.rel[cd.reloff + r] + 1];
            }
        }, 64);
        for (i64 i = 0; i < nrow_total; ++i) S.gth_ptr[i + 1] += S.gth_ptr[i];
        S.gth_src.assign(S.gth_ptr[nrow_total], 0);
        std::vector<i64> cur(S.gth_ptr.begin(), S.gth_ptr.end() - 1);
        parallel_for_throw(ns_total, nthreads, [&](unsigned, i64 s) {
            const FrontDesc &w = S.fronts[s];
            if (!S.front_local[s]) return;
            for (i32 t = 0; t < w.nchild; ++t) {
                const FrontDesc &cd = S.fronts[S.children[w.child_ptr + t]];
                const i32 rsc = cd.f - cd.ns;
                for (i32 r = 0; r < rsc; ++r) S.gth_src[cur[w.rowoff + S.rel[cd.reloff + r]]++] = cd.ucoff + r;
            }
        }, 64);
    }
    pt.mark("assembly lists");
    // ---- 14. assembly lists: S[ii,kk] = sum_j A[i,j] D_j A[k,j] (+ regD on the diagonal) ----
    {
        // (uvec: no zero-fill by resize; the defaults are written by the host threads, chunk by chunk)
        S.s_target.resize((size_t)S.nnzS); S.s_diag_row.resize((size_t)S.nnzS); S.s_local.resize((size_t)S.nnzS); S.pair_ptr.resize((size_t)S.nnzS + 1);
        {
            constexpr i64 CH = (i64)1 << 16;
            const i64 nch = (S.nnzS + CH) / CH;                  // covers pair_ptr[nnzS]
            if (!parallel_for(nch, host_threads(nch), [&](unsigned, i64 ch) {
                    const i64 e0 = ch * CH, e1 = std::min(S.nnzS, e0 + CH);
                    for (i64 e = e0; e < e1; ++e) { S.s_target[(size_t)e] = -1; S.s_diag_row[(size_t)e] = -1; S.s_local[(size_t)e] = 0; S.pair_ptr[(size_t)e] = 0; }
                    if (e1 == S.nnzS) S.pair_ptr[(size_t)S.nnzS] = 0;
                })) return fail(S, TLPK_OOM, "out of memory while building the assembly lists");
        }
        pt.mark("assembly lists: traverse");
        auto col_is_mine = [&](i32 j) -> bool {
            if (opt.nranks == 1 || !have_blocks) return true;
            const i32 b = col_block[j];
            return (b < 0) ? (opt.rank == 0) : (block_owner[b] == opt.rank);
        };
        // Round 6: ONE traversal of A per pivot column instead of two (count, then fill).  Fronts are independent (every stored entry of S belongs to exactly one
        // pivot column of exactly one front): handed out to the host threads.  A thread walks the products of a column once -- the order of the old fill pass:
        // rows of A' in order, entries of the column of A in order --, groups them by entry of S with a stable counting sort inside the column (a few hundred
        // products: cache-resident) and appends the column to its own stream; the counts go to pair_ptr.  After the prefix sum every column is ONE contiguous
        // copy from its thread's stream to its final place.  The lists are the lists of the two-pass version, entry by entry (tests/test_symbolic.py: digests).
        const unsigned nthreads = host_threads(ns_total);
        struct Stream { std::vector<double> w; std::vector<i32> j; std::vector<i32> le, cj, start; std::vector<double> cw; std::vector<i32> pos_in_front; std::vector<i64> epos; };
        std::vector<Stream> st(nthreads);
        uvec<i64> col_off((size_t)m);                     // per pivot column: offset of its products in the stream of ...
        uvec<i32> col_thr((size_t)m);                     // ... this thread (-1: a column of a front this rank does not own)
        {
            const bool ok = parallel_for(ns_total, nthreads, [&](unsigned tid, i64 s64) {
                const i32 s = (i32)s64;
                const FrontDesc &w = S.fronts[s];
                if (!S.front_local[s]) { for (i32 kk = w.col0; kk < w.col0 + w.ns; ++kk) col_thr[(size_t)kk] = -1; return; }
                Stream &T = st[tid];
                if (T.pos_in_front.empty()) { T.pos_in_front.assign(m, -1); T.epos.assign(m, -1); }
                std::vector<i32> &pos_in_front = T.pos_in_front;
                std::vector<i64> &epos = T.epos;
                const bool is_root = (s == S.root_front);
                for (i32 t = 0; t < w.f; ++t) pos_in_front[S.rowidx[w.rowoff + t]] = t;
                for (i32 kk = w.col0; kk < w.col0 + w.ns; ++kk) {
                    const i32 k = S.perm[kk];
                    const i64 e0 = S.Sp[kk], ne = S.Sp[kk + 1] - e0;
                    for (i64 e = e0; e < e0 + ne; ++e) {
                        const i32 ii = S.Si[e];
                        epos[ii] = e;
                        S.s_target[e] = w.loff + pos_in_front[ii] + pk_off(w.lda, kk - w.col0);
                        S.s_local[e] = 1;
                    }
                    if (opt.system == 1) { if (k >= opt.k2_n && (!is_root || opt.rank == 0)) S.s_diag_row[e0] = k - (i32)opt.k2_n; }   // constraint node: regD
                    else if (!is_root || opt.rank == 0) S.s_diag_row[e0] = k;
                    T.le.clear(); T.cw.clear(); T.cj.clear();
                    T.start.assign((size_t)ne + 1, 0);
                    auto emit = [&](i64 e, double wv, i32 j) { const i32 le = (i32)(e - e0); T.le.push_back(le); T.cw.push_back(wv); T.cj.push_back(j); ++T.start[(size_t)le + 1]; };
                    if (opt.system == 1) {
                        // Augmented system: the diagonal of a variable node is -(theta + regP) = -1 * D2[k]; an
                        // off-diagonal entry is the constant A[i,j] = A[i,j] * D2[k2_n] with D2[k2_n] = 1 (the columns
                        // of the incidence matrix carry (1, A[i,j]) on the variable / constraint node).
                        // Sharded runs: the assembled entries of the replicated root front (linking constraint nodes and the
                        // variable nodes of columns that touch linking rows only) belong to rank 0; the all-reduce of the root
                        // panel adds the ranks' extend-add contributions to them.
                        if (!(is_root && opt.rank != 0)) {
                            if (k < opt.k2_n) emit(e0, -1.0, k);
                            for (i64 q = S.Tp[k]; q < S.Tp[k + 1]; ++q) {
                                const i32 j = S.Tj[q];
                                const double akj = S.Ax[S.Tpos[q]];
                                for (i64 p = S.Ap[j]; p < S.Ap[j + 1]; ++p) {
                                    const i32 ii = S.iperm[S.Ai[p]];
                                    if (ii <= kk) continue;
                                    emit(epos[ii], akj * S.Ax[p], (i32)opt.k2_n);
                                }
                            }
                        }
                    } else {
                        for (i64 q = S.Tp[k]; q < S.Tp[k + 1]; ++q) {
                            const i32 j = S.Tj[q];
                            if (is_root && !col_is_mine(j)) continue;
                            const double akj = S.Ax[S.Tpos[q]];
                            for (i64 p = S.Ap[j]; p < S.Ap[j + 1]; ++p) {
                                const i32 ii = S.iperm[S.Ai[p]];
                                if (ii < kk) continue;
                                emit(epos[ii], akj * S.Ax[p], j);
                            }
                        }
                    }
                    // counts -> pair_ptr (prefix-summed below); stable counting sort of the column's products by entry into the thread's stream
                    const size_t np = T.le.size(), base = T.w.size();
                    for (i64 q = 0; q < ne; ++q) { S.pair_ptr[(size_t)(e0 + q) + 1] = T.start[(size_t)q + 1]; T.start[(size_t)q + 1] += T.start[(size_t)q]; }
                    T.w.resize(base + np); T.j.resize(base + np);
                    for (size_t q = 0; q < np; ++q) { const size_t at = base + (size_t)T.start[(size_t)T.le[q]]++; T.w[at] = T.cw[q]; T.j[at] = T.cj[q]; }
                    col_off[(size_t)kk] = (i64)base; col_thr[(size_t)kk] = (i32)tid;
                }
            }, 64);
            if (!ok) return fail(S, TLPK_OOM, "out of memory while building the assembly lists");
        }
        pt.mark("assembly lists: prefix");
        {
            // prefix sum in two levels: per chunk of 65 536 entries on the host threads, the chunk totals serially
            constexpr i64 CH = (i64)1 << 16;
            const i64 nch = (S.nnzS + CH - 1) / CH;
            std::vector<i64> tot((size_t)nch + 1, 0);
            bool ok = parallel_for(nch, host_threads(nch), [&](unsigned, i64 ch) {
                const i64 a = ch * CH, b = std::min(S.nnzS, a + CH);
                i64 acc = 0;
                for (i64 e = a; e < b; ++e) { acc += S.pair_ptr[(size_t)e + 1]; S.pair_ptr[(size_t)e + 1] = acc; }
                tot[(size_t)ch + 1] = acc;
            });
            for (i64 ch = 0; ch < nch; ++ch) tot[(size_t)ch + 1] += tot[(size_t)ch];
            ok = ok && parallel_for(nch, host_threads(nch), [&](unsigned, i64 ch) {
                const i64 a = ch * CH, b = std::min(S.nnzS, a + CH), add = tot[(size_t)ch];
                if (add) for (i64 e = a; e < b; ++e) S.pair_ptr[(size_t)e + 1] += add;
            });
            if (!ok) return fail(S, TLPK_OOM, "out of memory while building the assembly lists");
            S.pair_ptr[0] = 0;
        }
        pt.mark("assembly lists: place");
        {
            S.pair_w.resize((size_t)S.pair_ptr[S.nnzS]); S.pair_j.resize((size_t)S.pair_ptr[S.nnzS]);      // (first touched by the threads that fill them)
            const bool ok = parallel_for(m, host_threads(m / 256 + 1), [&](unsigned, i64 kk) {
                const i32 th = col_thr[(size_t)kk];
                if (th < 0) return;
                const i64 a = S.pair_ptr[(size_t)S.Sp[kk]], cnt = S.pair_ptr[(size_t)S.Sp[kk + 1]] - a;
                if (cnt <= 0) return;
                std::memcpy(S.pair_w.data() + a, st[(size_t)th].w.data() + col_off[(size_t)kk], (size_t)cnt * sizeof(double));
                std::memcpy(S.pair_j.data() + a, st[(size_t)th].j.data() + col_off[(size_t)kk], (size_t)cnt * sizeof(i32));
            }, 256);
            if (!ok) return fail(S, TLPK_OOM, "out of memory while building the assembly lists");
        }
    }
    pt.mark("schedule");
    S.error.clear();
    S.shared_device = opt.shared_device;
    build_schedule(S);
    if (!S.error.empty()) return TLPK_INTERNAL;
    // k_update walks its K range in slabs of 16 columns and relies on a slab lying inside ONE 64-column slice of the packed panel
    for (const UpdateTask &u : S.update_tasks)
        if ((u.k0 & 15) != 0) return fail(S, TLPK_INTERNAL, "update task: K range does not start on a multiple of 16 columns");
    pt.mark(nullptr);
    return TLPK_OK;
}

int analyse(Symbolic &S, i64 m, i64 n, const i64 *colptr, const i64 *rowval, const double *nzval, int base, const Options &opt) {
    const int rc = analyse_common(S, m, n, colptr, rowval, nzval, base, opt);
    return rc != TLPK_OK ? rc : analyse_rank(S, opt);
}

// ---------------------------------------------------------------------------------------------
// K2: the augmented system K = [-(Theta^-1 + Rp)  A'; A  Rd] of order N = n + m
// (/root/reference/src/KKT/KKT.jl:70-75, src/KKT/Cholmod/sqd.jl:5-74, src/KKT/LDLFactorizations/ldlfact.jl:63-139).
// K is symmetric quasi-definite: any symmetric permutation has a factorisation P K P' = L S L' with
// S = diag(+-1) known in advance (-1 for a variable node, +1 for a constraint node) and no pivoting
// (Vanderbei 1995), i.e. a "signed Cholesky" that runs on the same supernodal machinery.  The graph
// of K is the graph of B B' for the N x nnz(A) incidence matrix B whose column p holds 1 on the
// variable node and A[i,j] on the constraint node of the p-th nonzero of A: the whole analyse phase is
// reused on B, only the assembly lists differ (see step 14).
// ---------------------------------------------------------------------------------------------
// rank-independent part: incidence matrix, analyse_common on it, signs
int analyse_k2_common(Symbolic &S, i64 m, i64 n, const i64 *colptr, const i64 *rowval, const double *nzval,
                      int base, const Options &opt_in, Options *opt_out) {
    if (m < 0 || n < 0 || (base != 0 && base != 1) || !colptr) return fail(S, TLPK_BADARG, "bad dimensions or index base");
    if (opt_in.ordering == TLPK_ORDER_USER) return fail(S, TLPK_BADARG, "user_perm is not supported for K2");
    const i64 nnz = colptr[n] - base;
    if (nnz < 0 || m + n >= ((i64)1 << 31) || nnz >= ((i64)1 << 30)) return fail(S, TLPK_TOO_LARGE, "augmented system exceeds int32");
    if (nnz > 0 && (!rowval || !nzval)) return fail(S, TLPK_BADARG, "null rowval/nzval");
    std::vector<i64> bp((size_t)nnz + 1), bi((size_t)(2 * nnz));
    std::vector<double> bx((size_t)(2 * nnz));
    for (i64 j = 0; j < n; ++j) {
        if (colptr[j] - base < 0 || colptr[j + 1] < colptr[j] || colptr[j + 1] - base > nnz) return fail(S, TLPK_BADARG, "colptr not monotone");
        for (i64 p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
            const i64 r = rowval[p] - base;
            if (r < 0 || r >= m) return fail(S, TLPK_BADARG, "row index out of range");
            bp[(size_t)p] = 2 * p;
            bi[(size_t)(2 * p)] = j; bx[(size_t)(2 * p)] = 1.0;                  // variable node
            bi[(size_t)(2 * p + 1)] = n + r; bx[(size_t)(2 * p + 1)] = nzval[p];   // constraint node
        }
    }
    bp[(size_t)nnz] = 2 * nnz;
    Options opt = opt_in;
    opt.system = 1; opt.k2_n = n;
    // Block-angular LPs: the nodes of the augmented system inherit the blocks of the rows -- constraint node n + i that of
    // row i, variable node j that of the diagonal-block rows its column touches (a column with entries in linking rows
    // only, e.g. a linking row's slack, joins the linking nodes: -1 = root front).  Ordering per block, stream groups and
    // the root front then work as for K1; the root mixes both signs, which the signed Cholesky does not mind.
    std::vector<i64> node_block;
    if (opt_in.row_block) {
        node_block.assign((size_t)(m + n), -1);
        for (i64 i = 0; i < m; ++i) node_block[(size_t)(n + i)] = opt_in.row_block[i];
        for (i64 j = 0; j < n; ++j) {
            i64 b = -1;
            for (i64 p = colptr[j] - base; p < colptr[j + 1] - base; ++p) {
                const i64 rb = opt_in.row_block[rowval[p] - base];
                if (rb < 0) continue;
                if (b >= 0 && rb != b) return fail(S, TLPK_BADARG, "row_block is not a block-angular partition: a column couples two diagonal blocks");
                b = rb;
            }
            node_block[(size_t)j] = b;
        }
        opt.row_block = node_block.data();
    }
    const int rc = analyse_common(S, m + n, nnz, bp.data(), bi.data(), bx.data(), 0, opt);
    if (rc != TLPK_OK) return rc;
    S.system = 1; S.k2_n = n; S.k2_m = m;
    S.csign.resize((size_t)(m + n));
    for (i64 kk = 0; kk < m + n; ++kk) S.csign[(size_t)kk] = (S.perm[(size_t)kk] < n) ? -1.0 : 1.0;
    opt.row_block = nullptr;                    // (points into a local; analyse_rank reads the copy inside S)
    if (opt_out) *opt_out = opt;
    return TLPK_OK;
}

int analyse_k2(Symbolic &S, i64 m, i64 n, const i64 *colptr, const i64 *rowval, const double *nzval,
               int base, const Options &opt_in) {
    Options opt;
    const int rc = analyse_k2_common(S, m, n, colptr, rowval, nzval, base, opt_in, &opt);
    return rc != TLPK_OK ? rc : analyse_rank(S, opt);
}

// ---------------------------------------------------------------------------------------------
// Launch schedules.  All task lists are static for the lifetime of the handle: one IPM run
// replays them once per update! (factor) and 2..6 times per Newton step (solves).
// ---------------------------------------------------------------------------------------------
static void build_schedule(Symbolic &S) {
    PhaseTimer spt; spt.mark("schedule: prologue");
    // Scope of the level body being generated: stream group `cur_g` (fronts at depth >= 1 of that
    // group) or -1 = the depth-0 fronts, which run on the main stream after all groups joined.
    int cur_g = -1, cur_side = 0;
    std::vector<i64> region_slots;          // split-K scratch slots needed per (stream group, side) region
    // Isolated 1 x 1 fronts (an LP row that shares no column with any other row -- e.g. an inequality
    // row whose only entry is its slack: 13 % of the rows of the headline instance): one thread each in
    // k_single_factor / k_single_solve instead of a 256-thread workgroup in six different launches.
    S.front_single.assign(S.fronts.size(), 0);
    S.single_loff.clear(); S.single_dinvoff.clear(); S.single_col.clear();
    for (size_t s = 0; s < S.fronts.size(); ++s) {
        const FrontDesc &w = S.fronts[s];
        if (w.f == 1 && w.ns == 1 && w.nchild == 0 && w.parent < 0 && (i32)s != S.root_front) {
            S.front_single[s] = 1;
            if (S.front_local[s]) { S.single_loff.push_back(w.loff); S.single_dinvoff.push_back(w.dinvoff); S.single_col.push_back(w.col0); }
        }
    }
    // zero-fill of the panels before the assembly: per 64-column slice only the rows from the slice's first row down (the
    // blocks above the diagonal blocks are never read)
    // Step 13d (round 4): UPPER fronts.  On a block-angular LP 97 % of the factor's bytes are the panels of the diagonal blocks' top fronts (depth 1) and
    // the root, and nothing touches them before the extend-add of their level -- while the leaf levels below are a chain of short, latency-bound
    // launches that leave HBM idle.  Their zero-fill (0.9 ms of the 52 ms step on config C4, 2.4 of 137 ms on the north-star LP) and assembly therefore
    // run on a stream of their own beside the leaf levels; an LK_WAIT_UPPER marker makes a group's stream wait for them before its first launch
    // of an upper level.  Only with stream groups (the single-stream modes and graph replay keep the one-stream order).  TLPK_DEFER_UPPER=0: off.
    S.front_upper.assign(S.fronts.size(), 0);
    {
        static const bool defer = [] { const char *e = std::getenv("TLPK_DEFER_UPPER"); return !e || std::atoi(e) != 0; }();
        if (defer && S.ngroups >= 2 && S.nlevels >= 3)
            for (size_t s = 0; s < S.fronts.size(); ++s) {
                const FrontDesc &w = S.fronts[s];
                if (S.front_local[s] && !S.front_fa[s] && !S.front_single[s] && S.depth[s] <= 1 && (i64)w.lda * w.ns > 4096) S.front_upper[s] = 1;
            }
    }
    S.zero_tasks.clear(); S.zero_small.clear();
    for (int upper = 0; upper < 2; ++upper) {
        for (size_t s = 0; s < S.fronts.size(); ++s) {
            if (!S.front_local[s] || S.front_fa[s] || (int)S.front_upper[s] != upper) continue;       // (panels formed by k_front_assemble are written whole)
            const FrontDesc &w = S.fronts[s];
            if ((i64)w.lda * w.ns <= 4096) { S.zero_small.push_back((i32)s); continue; }      // whole panel by one wave
            for (i32 c0 = 0; c0 < w.ns; c0 += NB_IN) { S.zero_tasks.push_back((i32)s); S.zero_tasks.push_back(c0); }
        }
        if (!upper) S.n_zero_lower = (i64)S.zero_tasks.size() / 2;
    }
    auto in_scope = [&](i32 s) { return S.front_local[s] && !S.front_single[s] && (cur_g < 0 || S.front_group[s] == cur_g); };
    // structural-zero flags of a 128-row operand window [r0, r0 + TILE) of front s, one byte per K slab (step 13c), built on first use
    // (node-based map: the address of a flag vector stays valid while others are added -- a tile looks up two windows and keeps both pointers)
    std::vector<std::unordered_map<i32, std::vector<char>>> win_cache(S.fronts.size());
    // TLPK_SKIP_WIN (experiment): the window of rows a skip decision looks at, 128 (a tile's own rows) | 256 | 512: with a coarser window the tiles of a
    // super-tile skip the SAME slabs and keep walking K side by side (their operand loads meet in L2), at the price of fewer skipped slabs
    static const i32 skip_win = [] { const char *e = std::getenv("TLPK_SKIP_WIN"); const int v = e ? std::atoi(e) : TILE; return (v == 256 || v == 512) ? v : TILE; }();
    auto window_flags = [&](i32 s, i32 r0) -> const char * {            // r0 = first row of a tile (NOT always a multiple of TILE: the tiles of U start at row ns)
        auto &lst = win_cache[(size_t)s];
        { const auto it = lst.find(r0); if (it != lst.end()) return it->second.data(); }
        const FrontDesc &w = S.fronts[s];
        const i64 nsl = (w.ns + 15) / 16, W = ((w.f + 15) / 16 + 63) / 64;
        const uint64_t *bits = S.skip_bits.data() + S.skip_off[(size_t)s];
        std::vector<char> fl((size_t)nsl, 0);
        // rows looked at: the tile's own [r0, r0 + TILE), widened to whole skip_win-row windows when skip_win > TILE
        const i32 lo = (skip_win > TILE) ? r0 / skip_win * skip_win : r0;
        const i32 hi = (skip_win > TILE) ? (r0 + TILE + skip_win - 1) / skip_win * skip_win : r0 + TILE;
        const i32 g0 = lo / 16, g1 = (std::min(hi, w.f) - 1) / 16;
        for (i64 k = 0; k < nsl; ++k) {
            const uint64_t *b = bits + k * W;
            char any = 0;
            for (i32 g = g0; g <= g1 && !any; ++g) any = (char)((b[g >> 6] >> (g & 63)) & 1);
            fl[(size_t)k] = any;
        }
        return lst.emplace(r0, std::move(fl)).first->second.data();
    };
    auto push_launch = [&](std::vector<Launch> &L, i32 kind, i64 first, i64 count) {
        if (count > 0) L.push_back(Launch{kind, cur_g, first, count, cur_side, 0});
    };
    spt.mark("schedule: factor");
    // ---------------- factorisation ----------------
    auto factor_level = [&](i32 d) {
        const i32 t0 = S.level_ptr[d], t1 = S.level_ptr[d + 1];
        const bool root_level = (d == 0 && S.root_front >= 0);
        {
            bool any_upper = false;
            for (i32 t = t0; t < t1 && !any_upper; ++t) any_upper = in_scope(S.level_fronts[t]) && S.front_upper[(size_t)S.level_fronts[t]];
            if (any_upper) S.factor_launches.push_back(Launch{LK_WAIT_UPPER, cur_g, (i64)cur_g, 0, 0, 0});      // step 13d (`first` repeats the group: the exported triples carry no group)
        }
        // (a) extend-add, panel part: the children's update-matrix columns that land in the pivot
        // columns [0, ns) of their parent.  The U part [ns, f) is added AFTER the front's single
        // U update has written U (beta = 0), so U is never zero-filled nor read back by k_update.
        auto push_ea = [&](bool u_part) {
            const i64 first = (i64)S.ea_tasks.size();
            for (i32 t = t0; t < t1; ++t) {
                const i32 s = S.level_fronts[t];
                if (!in_scope(s)) continue;
                const FrontDesc &w = S.fronts[s];
                if (w.nchild == 0) continue;
                if (!u_part && S.front_fa[(size_t)s]) continue;      // panel part: k_front_assemble
                const i32 jbeg = u_part ? w.ns : 0, jend = u_part ? w.f : w.ns;
                const bool fa = S.front_fa[(size_t)s];
                const i32 cols = ea_cols(w, fa);
                i32 k = u_part ? ea_npan(w, fa) : 0;              // boundary index of j (section 13a)
                // TLPK_EA_BANDS (experiment): the rows of a big parent are cut into bands of whole boundary ranges, one workgroup per (column range, band)
                static const i32 nbands = [] { const char *e = std::getenv("TLPK_EA_BANDS"); return e ? std::max(1, std::atoi(e)) : 1; }();
                const i32 nbnd = ea_nbounds(w, fa);
                const i32 bands = (w.f >= 2048) ? nbands : 1;
                for (i32 j = jbeg; j < jend; j += cols, ++k) {
                    if (bands == 1) { S.ea_tasks.push_back(EaTask{s, j, std::min(j + cols, jend), k, 0, 0, 0, 0}); continue; }
                    // rows >= j only matter (lower triangle): bands over the boundaries [k, nbnd - 1)
                    const i32 span = nbnd - 1 - k, per = (span + bands - 1) / bands;
                    for (i32 b0 = k; b0 < nbnd - 1; b0 += std::max(per, 1)) S.ea_tasks.push_back(EaTask{s, j, std::min(j + cols, jend), k, b0, std::min(b0 + std::max(per, 1), nbnd - 1), 0, 0});
                }
            }
            push_launch(S.factor_launches, LK_EXTEND_ADD, first, (i64)S.ea_tasks.size() - first);
        };
        {
            // panels of the large fronts: tiles of FA_CW columns x <= 256 rows, every stored entry of the panel written exactly once
            // (rows from the first row of the column tile's 64-column slice down: what the packed panel stores)
            const i64 first = (i64)S.fa_tasks.size();
            for (i32 t = t0; t < t1; ++t) {
                const i32 s = S.level_fronts[t];
                if (!in_scope(s) || !S.front_fa[(size_t)s]) continue;
                const FrontDesc &w = S.fronts[s];
                const i32 npan = ea_npan(w, true), nbnd = ea_nbounds(w, true);
                for (i32 bc = 0; bc < npan; ++bc) {
                    const i32 j0 = bc * FA_CW;
                    for (i32 br0 = ((j0 >> 6) << 6) / FA_CW; br0 < nbnd - 1; br0 += FA_RB)
                        S.fa_tasks.push_back(FaTask{s, bc, br0, std::min(br0 + FA_RB, nbnd - 1)});
                }
            }
            push_launch(S.factor_launches, LK_FRONT_ASSEMBLE, first, (i64)S.fa_tasks.size() - first);
        }
        push_ea(false);
        if (root_level) S.factor_launches.push_back(Launch{LK_ALLREDUCE_ROOT, -1, 0, 0});
        // (b) blocked partial factorisation.  Outer level LEFT-looking: before the 256-wide block
        // column `io` of a front is factorised, one MFMA update accumulates the contribution of
        // ALL previous columns [0, ko) in registers and writes each target entry once (the
        // right-looking variant re-wrote the whole trailing matrix every 256 columns and was
        // HBM-bound on that read-modify-write).  The update matrix U gets a single update with
        // K = [0, ns) after the last block column.
        i32 max_ns = 0;
        for (i32 t = t0; t < t1; ++t) if (in_scope(S.level_fronts[t])) max_ns = std::max(max_ns, S.fronts[S.level_fronts[t]].ns);
        const i32 nouter = (max_ns + NB_OUT - 1) / NB_OUT;
        // part: 0 = only the tiles of the block column's diagonal block (rows < c0 + NB_OUT),
        //       1 = only the tiles below it, 2 = all
        // dry != nullptr: only count the tiles (into *dry), for every front the caller passes
        i64 *dry = nullptr;
        i32 upd_super = 4;
        if (const char *e = std::getenv("TLPK_UPD_SUPER")) upd_super = std::max(1, std::atoi(e));   // tuning knob
        // Canonical index of a tile inside its launch: position in the list that ALL of this rank's fronts of the level would
        // produce (level order), whatever the stream group the front runs in -- what the tail split below is decided on.
        std::vector<i64> canon_count(S.fronts.size(), 0), canon_next(S.fronts.size(), 0);
        std::vector<i64> task_canon;                      // canonical index of every task pushed by the current launch
        bool allow_skip = true;                           // off for split-K launches (few tiles: the parts are cut by K position)
        std::vector<char> need_tmp;
        // Round 6: the fronts of the level with more than one block column may run as ONE dependency-driven launch (LK_CHAIN, below): pass 0 = every front
        // through the launches; pass 1 = the other fronts through the launches, pass 2 = the chain fronts, their launches CAPTURED and turned into items.
        // The decisions that look at the whole level (split-K, macro columns, look-ahead) see all of the rank's fronts in every pass: a tile is the same
        // tile whichever way it is launched.
        int pass = 0;
        // TLPK_CHAIN_TILE64 (diagnostics): 0 = the diagonal block's short update as 128 x 128 tiles, 1 = 64 x 64 tiles (update_tile64), unset / 2 = 32 x 32 tiles (update_tile32)
        const i32 chain_tile = [] { const char *e = std::getenv("TLPK_CHAIN_TILE64"); const int v = e ? std::atoi(e) : 2; return v == 0 ? TILE : (v == 1 ? 64 : 32); }();
        std::vector<char> chain_front(S.fronts.size(), 0);       // (only the entries of this level's fronts are ever set)
        struct Cap { i32 kind; i64 first, count; };
        std::vector<Cap> cap;
        i64 chain_slot_base = 0;                                 // split-K scratch slots of a chain launch are never reused inside the launch
        auto pass_ok = [&](i32 s) { return pass == 0 || ((bool)chain_front[(size_t)s] == (pass == 2)); };
        auto emit = [&](i32 kind, i64 first, i64 count) {
            if (count <= 0) return;
            if (pass == 2) cap.push_back(Cap{kind, first, count}); else push_launch(S.factor_launches, kind, first, count);
        };
        // entries of a tile that are targets: row >= column, row < f, column < c1
        auto tile_entries = [&](const FrontDesc &w, i32 i0, i32 j0, i32 c1, i32 ts = TILE) {
            double e = 0;
            const i32 r1 = std::min(i0 + ts, w.f);
            for (i32 col = j0; col < std::min(j0 + ts, c1); ++col) e += std::max(0, r1 - std::max(i0, col));
            return e;
        };
        // ts = 64 / 32 (chain launches only, part 0, K <= 256): the diagonal block's tiles as 64 x 64 tiles (UpdateTask.pad2 = 1, kernels.hip: update_tile64) or
        // 32 x 32 tiles (pad2 = 2, update_tile32: one 16 x 16 block per wave, operands straight from the panel) -- the short update that is left on the chain
        // behind a solved block column runs on ten / 36 CUs instead of three.  Same sums in the same order per entry.
        auto push_update_region = [&](i32 s, const FrontDesc &w, i32 k0, i32 kw, i32 c0, i32 c1, i32 beta0, int part, i32 ts = TILE) {
            if (kw <= 0 || c0 >= c1) return;
            if (ts < TILE) {
                for (i32 j0 = c0; j0 < c1; j0 += ts)
                    for (i32 i0 = j0; i0 < std::min(c0 + NB_OUT, w.f); i0 += ts) {
                        i32 seg = 0, nsl = 0;
                        double kexec = kw;
                        const i32 nfull = kw / 16;
                        if (allow_skip && S.skip_off[(size_t)s] >= 0 && nfull >= 2) {
                            const char *fi = window_flags(s, i0), *fj = window_flags(s, j0);      // (the 128-row windows that hold the tile's rows: never skips a needed slab)
                            const i32 sl0 = k0 / 16;
                            i32 cnt = 0;
                            for (i32 k = 0; k < nfull; ++k) cnt += (fi[sl0 + k] & fj[sl0 + k]);
                            if (cnt == 0 && !beta0 && kw % 16 == 0) { if (!dry) S.flops_update_skipped += 2.0 * kw * tile_entries(w, i0, j0, c1, ts); continue; }
                            if (cnt < nfull) {
                                need_tmp.assign((size_t)nfull, 0);
                                for (i32 k = 0; k < nfull; ++k) need_tmp[(size_t)k] = fi[sl0 + k] & fj[sl0 + k];
                                nsl = cnt; kexec = 16.0 * cnt + kw % 16;
                                if (!dry) {
                                    seg = (i32)S.upd_seg.size() + 1;
                                    S.upd_seg.push_back(0);
                                    i32 nseg = 0;
                                    for (i32 k = 0; k < nfull;) {
                                        if (!need_tmp[(size_t)k]) { ++k; continue; }
                                        i32 e = k; while (e < nfull && need_tmp[(size_t)e]) ++e;
                                        S.upd_seg.push_back(k0 + 16 * k); S.upd_seg.push_back(e - k); ++nseg;
                                        k = e;
                                    }
                                    S.upd_seg[(size_t)seg - 1] = nseg;
                                }
                            }
                        }
                        if (dry) { ++*dry; ++canon_count[(size_t)s]; }
                        else {
                            const double ent = tile_entries(w, i0, j0, c1, ts);
                            S.flops_update += 2.0 * kexec * ent; S.flops_update_skipped += 2.0 * (kw - kexec) * ent;
                            if (pass == 2) S.flops_update_chain += 2.0 * kexec * ent;
                            S.update_tasks.push_back(UpdateTask{s, k0, kw, i0, j0, c1, beta0, 0, seg, nsl, ts == 64 ? 1 : 2, 0}); task_canon.push_back(canon_next[(size_t)s]++);
                        }
                    }
                return;
            }
            // tiles in super-tile order (UPD_SUPER x UPD_SUPER tiles): tasks that are neighbours in the list
            // read the same row / column slabs of the panel, and k_update deals runs of 64 consecutive
            // tasks to one XCD (one L2)
            const i32 SUP = upd_super * TILE;
            for (i32 J0 = c0; J0 < c1; J0 += SUP)
                for (i32 I0 = J0; I0 < w.f; I0 += SUP)
                    for (i32 j0 = J0; j0 < std::min(J0 + SUP, c1); j0 += TILE)
                        for (i32 i0 = std::max(I0, j0); i0 < std::min(I0 + SUP, w.f); i0 += TILE) {
                            const bool diag_blk = i0 < c0 + NB_OUT;
                            if ((part == 0 && !diag_blk) || (part == 1 && diag_blk)) continue;
                            // K slabs in which both operand row ranges of the tile have a structural nonzero (step 13c)
                            i32 seg = 0, nsl = 0;
                            double kexec = kw;
                            const i32 nfull = kw / 16;
                            if (allow_skip && S.skip_off[(size_t)s] >= 0 && nfull >= 2) {
                                const char *fi = window_flags(s, i0), *fj = window_flags(s, j0);
                                const i32 sl0 = k0 / 16;
                                i32 cnt = 0;
                                for (i32 k = 0; k < nfull; ++k) cnt += (fi[sl0 + k] & fj[sl0 + k]);
                                if (cnt == 0 && !beta0 && kw % 16 == 0) { if (!dry) S.flops_update_skipped += 2.0 * kw * tile_entries(w, i0, j0, c1); continue; }   // the tile receives nothing
                                if (cnt < nfull) {
                                    need_tmp.assign((size_t)nfull, 0);
                                    for (i32 k = 0; k < nfull; ++k) need_tmp[(size_t)k] = fi[sl0 + k] & fj[sl0 + k];
                                    for (i32 k = nfull - 1; k >= 0 && cnt < 2; --k) if (!need_tmp[(size_t)k]) { need_tmp[(size_t)k] = 1; ++cnt; }   // the kernel's pipeline wants >= 2 slabs
                                    nsl = cnt; kexec = 16.0 * cnt + kw % 16;
                                    if (!dry) {
                                        seg = (i32)S.upd_seg.size() + 1;
                                        S.upd_seg.push_back(0);
                                        i32 nseg = 0;
                                        for (i32 k = 0; k < nfull;) {
                                            if (!need_tmp[(size_t)k]) { ++k; continue; }
                                            i32 e = k; while (e < nfull && need_tmp[(size_t)e]) ++e;
                                            S.upd_seg.push_back(k0 + 16 * k); S.upd_seg.push_back(e - k); ++nseg;
                                            k = e;
                                        }
                                        S.upd_seg[(size_t)seg - 1] = nseg;
                                    }
                                }
                            }
                            if (dry) { ++*dry; ++canon_count[(size_t)s]; }
                            else {
                                const double ent = tile_entries(w, i0, j0, c1);
                                S.flops_update += 2.0 * kexec * ent; S.flops_update_skipped += 2.0 * (kw - kexec) * ent;
                                if (pass == 2) S.flops_update_chain += 2.0 * kexec * ent;
                                S.update_tasks.push_back(UpdateTask{s, k0, kw, i0, j0, c1, beta0, 0, seg, nsl}); task_canon.push_back(canon_next[(size_t)s]++);
                            }
                        }
        };
        auto for_fronts = [&](auto &&fn) {              // dry runs see all of the rank's fronts of the level
            for (i32 t = t0; t < t1; ++t) {
                const i32 s = S.level_fronts[t];
                if (dry ? (bool)S.front_local[s] : (in_scope(s) && pass_ok(s))) fn(s, S.fronts[s]);
            }
        };
        // One update launch: `gen` pushes its tiles.  Split-K: when the launch would leave most of the
        // chip idle (fewer than ~128 tiles over ALL of the rank's fronts of the level -- the decision
        // must not depend on the stream groups), the K range of every tile is cut into up to 8 parts of
        // >= 256 columns, computed by different workgroups into scratch and applied in order by a
        // k_update_reduce launch.  A tile with K = 3300 runs for ~0.9 ms whatever runs beside it: with
        // 8 blocks per rank (8-GPU sharding), for the root front, and for the diagonal-block tiles on
        // the side stream this is the critical path.
        // Tail split (round 3, OFF by default: measured without gain).  The tiles of a launch have the same K, i.e. the same
        // duration T, and the chip holds 512 of them at a time (2 workgroups x 256 CUs): on paper a launch of 2.4 x 512 tiles takes
        // 3 T, the last T with 60 % of the slots empty, and cutting the r = (tiles mod slots) tiles of the last wave along K into
        // p parts (p minimising ceil(r p / slots) / p) lifts the simulated slot efficiency of the C4 schedule from 0.84 to 0.98.
        // On the device the update time did not move (32.1 -> 32.5 ms + 0.75 ms of reductions; profiles/r03_tail_split.txt): a
        // workgroup that has its CU to itself runs at nearly twice the rate of two sharing the matrix pipes, so a half-empty last
        // wave is not half idle.  TLPK_TAIL_SLOTS=512 turns the split on (tiles are chosen by their canonical index over ALL of
        // the rank's fronts of the level, so that results do not depend on the number of stream groups).
        // (the look-ahead rule: commented where the block columns are laid out, below)
        const int la_env = [] { const char *e = std::getenv("TLPK_LOOKAHEAD"); return e ? (std::atoi(e) != 0 ? 1 : 0) : -1; }();
        bool lookahead = la_env == 1;
        if (la_env < 0) {
            i32 nbig = 0, ns_max = 0;
            for (i32 t = t0; t < t1; ++t) {
                const i32 sf = S.level_fronts[t];
                if (!S.front_local[sf] || S.front_single[sf]) continue;
                if (S.fronts[sf].ns > NB_OUT) { ++nbig; ns_max = std::max(ns_max, S.fronts[sf].ns); }
            }
            lookahead = nbig >= 1 && nbig <= 16 && ns_max <= 12288;
        }
        i64 UPD_SLOTS = 0;
        if (const char *e = std::getenv("TLPK_TAIL_SLOTS")) UPD_SLOTS = std::atoll(e);       // tuning knob; 0 = no tail split
        i64 TAIL64 = 0;                                                                        // last round of an update launch as 64 x 64 tiles when it holds at most this many tiles (0 = off: MEASURED SLOWER, see below)
        if (const char *e = std::getenv("TLPK_TAIL64")) TAIL64 = std::max(0, std::atoi(e));   // tuning knob
        i64 TAIL64_SLOTS = 512;                                                                // resident 128 x 128 tiles (2 workgroups x 256 CUs); TLPK_TAIL64_SLOTS: for the CPU tests of the tail shape on small LPs
        if (const char *e = std::getenv("TLPK_TAIL64_SLOTS")) TAIL64_SLOTS = std::max(1, std::atoi(e));
        i32 KSPLIT_LEN = 0;                                                                  // look-ahead levels: longest K range of one update item (0 = off)
        if (const char *e = std::getenv("TLPK_KSPLIT_LEN")) KSPLIT_LEN = std::max(0, std::atoi(e));       // tuning knob (multiples of 16 keep the parts on slab boundaries)
        auto emit_update_launch = [&](auto &&gen) {
            i64 t_level = 0;
            for (i32 t = t0; t < t1; ++t) canon_count[(size_t)S.level_fronts[t]] = 0;
            allow_skip = true;
            dry = &t_level; gen(); dry = nullptr;
            i64 want = 256;
            if (const char *e = std::getenv("TLPK_SPLITK_TILES")) want = std::atoll(e);       // tuning knob; 0 = off
            i32 nsplit = (t_level > 0) ? (i32)std::min<i64>(8, want / t_level) : 1;
            if (nsplit >= 2 || UPD_SLOTS > 0) {           // split-K launches cut the K range by position: no skip lists there
                allow_skip = false;
                for (i32 t = t0; t < t1; ++t) canon_count[(size_t)S.level_fronts[t]] = 0;
                t_level = 0; dry = &t_level; gen(); dry = nullptr;
                nsplit = (t_level > 0) ? (i32)std::min<i64>(8, want / t_level) : 1;
            }
            { i64 acc = 0; for (i32 t = t0; t < t1; ++t) { const i32 s = S.level_fronts[t]; canon_next[(size_t)s] = acc; acc += canon_count[(size_t)s]; } }
            const i64 f_upd = (i64)S.update_tasks.size();
            task_canon.clear();
            gen();
            // Round 6, look-ahead levels (the levels the dependency-driven launch serves): NO item may run for longer than a link of the chain.  A tile with
            // K = 4096 holds its workgroup for 640 us -- four block columns of the chain -- and the block column that waits for it (its strips) stalls that long
            // whatever the number of tiles beside it.  Every tile of such a level is cut by K LENGTH, into parts of at most KSPLIT_LEN columns (<= 8 parts),
            // whatever the number of tiles in the launch; the reduction applies the parts in order.  (Split tiles take no skip lists: regenerate without them.)
            bool ksplit = false;
            if (lookahead && KSPLIT_LEN > 0 && nsplit < 2 && UPD_SLOTS == 0) {
                for (i64 q = f_upd; q < (i64)S.update_tasks.size(); ++q) if (S.update_tasks[(size_t)q].kw > KSPLIT_LEN && !S.update_tasks[(size_t)q].pad2) { ksplit = true; break; }
                if (ksplit) {
                    S.update_tasks.resize((size_t)f_upd);
                    task_canon.clear();
                    { i64 acc = 0; for (i32 t = t0; t < t1; ++t) { const i32 s = S.level_fronts[t]; canon_next[(size_t)s] = acc; acc += canon_count[(size_t)s]; } }
                    allow_skip = false;
                    gen();
                }
            }
            allow_skip = true;
            const i64 cnt = (i64)S.update_tasks.size() - f_upd;
            if (cnt == 0) return;
            // Longest first (TLPK_UPD_LPT, round 4): tiles that skip K slabs are shorter than their neighbours; dealt out last they fill the tail of
            // the launch instead of leaving long tiles to finish alone.  Stable: tiles of equal length keep the super-tile order (L2 locality).
            static const bool lpt = [] { const char *e = std::getenv("TLPK_UPD_LPT"); return e && std::atoi(e) != 0; }();
            if (lpt && nsplit < 2 && UPD_SLOTS == 0) {
                auto len = [](const UpdateTask &t) { return t.seg ? 16 * t.nsl + t.kw % 16 : t.kw; };
                std::vector<size_t> idx((size_t)cnt);
                std::iota(idx.begin(), idx.end(), 0);
                std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return len(S.update_tasks[(size_t)f_upd + a]) > len(S.update_tasks[(size_t)f_upd + b]); });
                std::vector<UpdateTask> tmp_t((size_t)cnt); std::vector<i64> tmp_c((size_t)cnt);
                for (size_t q = 0; q < (size_t)cnt; ++q) { tmp_t[q] = S.update_tasks[(size_t)f_upd + idx[q]]; tmp_c[q] = task_canon[idx[q]]; }
                std::copy(tmp_t.begin(), tmp_t.end(), S.update_tasks.begin() + f_upd);
                task_canon.swap(tmp_c);
            }
            i64 tail_from = t_level; i32 tail_parts = 1;      // tiles with canonical index >= tail_from are cut into tail_parts
            // number of parts p in 1..8 that minimises the time of a wave of r equal tiles on UPD_SLOTS slots: ceil(r p / slots) / p
            auto best_parts = [&](i64 r) {
                i32 best = 1; double tbest = (double)((r + UPD_SLOTS - 1) / UPD_SLOTS);
                for (i32 p = 2; p <= 8; ++p) {
                    const double tp = (double)((r * p + UPD_SLOTS - 1) / UPD_SLOTS) / p;
                    if (tp < tbest - 1e-9) { tbest = tp; best = p; }
                }
                return best;
            };
            if (nsplit < 2 && UPD_SLOTS > 0 && t_level > 0) {
                const i64 r = t_level % UPD_SLOTS;
                if (t_level < UPD_SLOTS) nsplit = best_parts(t_level);                    // a single, partly filled wave: cut every tile
                else if (r > 0) { tail_parts = best_parts(r); tail_from = t_level - r; }  // the last wave
            }
            if (nsplit < 2 && tail_parts < 2 && !ksplit) {
                // Round 6 (the review's tail shape): the chip holds 512 of a launch's 128 x 128 tiles at a time, and the r = tiles mod 512 tiles of the last round
                // take a whole round -- 18 % of the serialised update time of config C4 (tools/update_launch_eff.py: 5.96 of 32.96 ms).  When r is small the last
                // round's tiles are cut into their 64 x 64 quarters (UpdateTask.pad2 = 1, update_tile64: four waves per workgroup, four workgroups per CU): 4 r
                // quarter-length items on 1024 slots, launched right behind the full rounds.  Same K ranges / segment lists, every entry sums its K columns in the
                // same order: same bits (CPU emulator and device: tests/test_symbolic.py, tests/test_gpu_parity.py).  MEASURED (profiles/r06_tail64.txt) and OFF by default
                // (TLPK_TAIL64=288 turns it on): C4 52.5 vs 51.5 ms per step, north-star 138.0 vs 136.6, and the serialised `roofline.frac` FALLS (0.633 vs 0.638, north-star
                // 0.598 vs 0.620): a 128 x 128 tile that has its CU to itself in a half-empty last round runs at nearly twice the rate of two sharing the matrix pipes, the
                // four quarter tiles re-read the operands and pay a launch boundary.  Not for the side stream's diagonal tiles (few, and the single-stream modes merge them with the rows-below launch by task
                // range), not inside the dependency-driven launches (pass 2: their items are already finer), not for launches of less than one round.
                i64 r = (TAIL64 > 0 && pass != 2 && cur_side == 0 && cnt >= TAIL64_SLOTS) ? cnt % TAIL64_SLOTS : 0;
                if (r > TAIL64) r = 0;
                if (r > 0) {
                    std::vector<UpdateTask> tail(S.update_tasks.end() - r, S.update_tasks.end());
                    S.update_tasks.resize(S.update_tasks.size() - (size_t)r);
                    const i64 f_t64 = (i64)S.update_tasks.size();
                    for (const UpdateTask &t : tail) {
                        const FrontDesc &w = S.fronts[t.front];
                        for (i32 dj = 0; dj < TILE; dj += 64)
                            for (i32 di = 0; di < TILE; di += 64) {
                                const i32 si = t.i0 + di, sj = t.j0 + dj;
                                if (si >= w.f || sj >= t.jlim || si + 63 < sj) continue;      // outside the front / the column range / above the diagonal
                                S.update_tasks.push_back(UpdateTask{t.front, t.k0, t.kw, si, sj, t.jlim, t.beta0, 0, t.seg, t.nsl, 1, 0});
                            }
                    }
                    emit(LK_UPDATE, f_upd, cnt - r);
                    emit(LK_UPDATE_T64, f_t64, (i64)S.update_tasks.size() - f_t64);
                    return;
                }
                emit(LK_UPDATE, f_upd, cnt); return;
            }
            std::vector<UpdateTask> orig(S.update_tasks.begin() + f_upd, S.update_tasks.end());
            S.update_tasks.resize(f_upd);
            const i64 f_red = (i64)S.reduce_tasks.size();
            i32 slot = (pass == 2) ? (i32)chain_slot_base : 0;
            for (size_t q = 0; q < orig.size(); ++q) {
                const UpdateTask &t = orig[q];
                const i32 limit = (nsplit >= 2) ? nsplit : (task_canon[q] >= tail_from ? tail_parts : 1);
                const i32 parts = ksplit ? (t.pad2 ? 1 : std::min<i32>(8, (t.kw + KSPLIT_LEN - 1) / KSPLIT_LEN)) : std::min(limit, t.kw / 256);
                if (parts < 2) { S.update_tasks.push_back(t); continue; }
                const i32 base = (t.kw / parts) / 16 * 16;              // multiples of the kernel's K slab
                i32 k = 0;
                for (i32 sp = 0; sp < parts; ++sp) {
                    const i32 kw_s = (sp == parts - 1) ? t.kw - k : base;
                    S.update_tasks.push_back(UpdateTask{t.front, t.k0 + k, kw_s, t.i0, t.j0, t.jlim, t.beta0, slot + sp + 1});
                    k += kw_s;
                }
                S.reduce_tasks.push_back(UpdateTask{t.front, slot, parts, t.i0, t.j0, t.jlim, t.beta0, 0});
                slot += parts;
            }
            // slots are relative to the scratch region of this launch's stream for now (see below)
            const int region = (cur_g + 1) * 2 + cur_side;
            if ((int)region_slots.size() <= region) region_slots.resize(region + 1, 0);
            region_slots[region] = std::max<i64>(region_slots[region], slot);
            if (pass == 2) chain_slot_base = slot;
            emit(LK_UPDATE, f_upd, (i64)S.update_tasks.size() - f_upd);
            emit(LK_UPDATE_REDUCE, f_red, (i64)S.reduce_tasks.size() - f_red);
        };
        // Macro columns: G consecutive block columns share ONE left-looking update with
        // K = [0, kM) (kM = first column of the macro column); inside the macro column a block
        // column only adds the short update K = [kM, ko).  G is chosen at the start of every macro
        // column so that the long-K launch has >= ~2000 tiles (4 waves of the chip): G = 1 (every
        // block column pulls all previous columns itself) while a block-column launch is that big
        // anyway; a level with a single huge front (general sparse LPs), and the last block columns
        // of any level, get wider macro columns -- a tile with K = 40 000 runs for 10 ms whatever
        // the number of tiles beside it.  G depends on all of this rank's fronts of the level, not only
        // on the current stream group's: the rounding must not depend on the number of streams
        // (across rank counts the all-reduce order differs anyway; a rank with few blocks needs the
        // wider macro columns to fill its GPU).
        constexpr i32 G_MAX = 16;
        i64 TILES_WANTED = 2048;
        if (const char *e = std::getenv("TLPK_MACRO_TILES")) TILES_WANTED = std::atoll(e);    // tuning knob; 0 = no macro columns
        const bool la_full = [] { const char *e = std::getenv("TLPK_LA_FULL"); return e && std::atoi(e) != 0; }();
        const bool la_macro = [] { const char *e = std::getenv("TLPK_LA_MACRO"); return !e || std::atoi(e) != 0; }();
        auto macro_width = [&](i32 ko) {
            i64 tiles_bc = 0;
            for (i32 t = t0; t < t1; ++t) {
                if (!S.front_local[S.level_fronts[t]]) continue;
                const FrontDesc &w = S.fronts[S.level_fronts[t]];
                if (w.ns > ko + NB_OUT) tiles_bc += 2 * (i64)((w.f - ko + TILE - 1) / TILE);
            }
            // launches of >= ~1000 tiles are left alone (measured on C4: macro columns there cost 0.3 ms,
            // two stream groups already fill each other's tails)
            if (tiles_bc <= 0 || 2 * tiles_bc >= TILES_WANTED) return (i32)1;
            // (TLPK_LA_MACRO=0, diagnostics: no macro columns on the look-ahead levels -- every block column pulls K = [0, ko - 256) one block column early.
            // Measured WORSE, pds-class 13.8 -> 14.6 ms: in the middle of a 7 900-column front a block column's update is 130 us of the whole chip, as long as a
            // link of the chain; the macro columns do that work early, while the chain is latency-bound, the pure left-looking form does it when it is due.)
            if (lookahead && !la_macro) return (i32)1;
            return (i32)std::min<i64>(G_MAX, (TILES_WANTED + tiles_bc - 1) / tiles_bc);
        };
        // macro column of every block column: block columns [mac_first[io], mac_first[io] + mac_G[io])
        std::vector<i32> mac_first((size_t)nouter + 2, 0), mac_G((size_t)nouter + 2, 1);
        {
            i32 G = 1, io_macro = 0;
            for (i32 io = 0; io <= nouter + 1; ++io) {
                if (io >= io_macro + G) { io_macro = io; G = macro_width(io * NB_OUT); }
                else if (io == 0) G = macro_width(0);
                mac_first[(size_t)io] = io_macro; mac_G[(size_t)io] = G;
            }
        }
        auto k_first = [&](i32 io) { return (io == mac_first[(size_t)io]) ? 0 : mac_first[(size_t)io] * NB_OUT; };   // block column io still needs K = [k_first, ko)
        // Look-ahead for the diagonal blocks (round 4).  The chain potrf(io) -> trsm(io) -> [update of the diagonal block of io + 1] -> potrf(io + 1) is
        // the critical path of a level with one big front (pds-class LPs: 31 block columns), of the root front, and of every rank of a sharded job.
        // The left-looking update of that diagonal block had K = [0, ko + 256): a few tiles with K up to the whole front, cut by split-K and followed by a
        // reduction -- 0.15 .. 0.3 ms on the chain per block column.  Now the part K = [k_first, ko) (everything but the block column just finished) rides in
        // the rows-below launch of block column io (same operands, same readiness: block columns < io), off the chain; behind trsm(io) only
        // K = [ko, ko + 256) is left: 3 tiles x 16 slabs.
        // MEASURED (profiles/r04_lookahead.txt) and OFF by default (TLPK_LOOKAHEAD=1 turns it on): pds-class LP 18.86 -> 18.5 ms per step -- split-K had already
        // cut the long-K diagonal update to ~0.1 ms and the chain is the potrf kernel itself (7.8 of 13.8 ms) --, C4 / north-star LP unchanged, and the C3
        // shape LOSES 11 % (675 -> 750 ms: inside its 16-wide macro columns the look-ahead tiles are a second long-K tail in every rows-below launch).
        // Round 5: AUTO (TLPK_LOOKAHEAD unset) turns it on for the levels it was measured to help -- at most 16 of this rank's fronts have more than
        // one block column and none has more than 12 288 pivot columns (a pds-class top front, the root front, the few blocks of a rank of an
        // 8-GPU job; not the C3 shape's 48 000-column front, not the 32 blocks per stream group of config C4 at N = 1, whose diagonal-block chains
        // are hidden behind the bulk updates anyway).  The rule looks at the rank's fronts of the level only, never at the stream groups.
        auto block_columns = [&]() {
        i32 pmax = 0;
        for (i32 t = t0; t < t1; ++t) if (in_scope(S.level_fronts[t]) && pass_ok(S.level_fronts[t])) pmax = std::max(pmax, S.fronts[S.level_fronts[t]].ns);
        if (pmax == 0 && pass != 0) return;
        const i32 nouter = (pass == 0) ? (max_ns + NB_OUT - 1) / NB_OUT : (pmax + NB_OUT - 1) / NB_OUT;      // (shadows the level's: block columns of THIS pass's fronts)
        for (i32 io = 0; io <= nouter; ++io) {
            const i32 ko = io * NB_OUT;
            const i32 io_macro = mac_first[(size_t)io], G = mac_G[(size_t)io];
            const i32 gi = io - io_macro, kM = io_macro * NB_OUT;
            // Round 6, with the look-ahead: the FIRST block column of a macro column (gi == 0) used to pull all of K = [0, ko) itself -- tiles of up to 4096 columns
            // (450 - 600 us each) between strips(io - 1) and strips(io), ON the chain (profiles/r06_chain_trace_pds.txt: one 450 us stall per macro column start).
            // Now K = [0, ko - 256) comes with block column io - 1 (same operands, ready one block column earlier, off the chain) for the whole block column, not
            // only for its diagonal block; behind strips(io - 1) only the short K = [ko - 256, ko) is left, as for every other block column.
            // TLPK_LA_FULL=1 (diagnostics): the same for EVERY block column of a look-ahead level (inside the first macro column K = [0, ko) grows with ko).
            const bool mac_la = lookahead && io >= 2 && (gi == 0 || la_full);
            const i32 ka_base = (gi == 0) ? 0 : kM;
            const i32 ka = mac_la ? std::max(ka_base, ko - NB_OUT) : ka_base;       // this block column still needs K = [ka, ko)
            const i32 kd = lookahead ? std::max(ka, ko - NB_OUT) : ka;      // ... its diagonal block only K = [kd, ko): the rest came with block column io - 1
            // Block column io.  The left-looking update of its DIAGONAL block and the factorisation
            // of that block (k_potrf*: a serial chain inside one workgroup per front) go to the
            // group's side stream; the update of the rows below runs concurrently on the group's
            // stream and hides them.  Only stream order and events: correct under any scheduling
            // (a profiler that serialises dispatches included).
            const bool overlap = io > 0 && io < nouter;
            if (overlap && pass != 2) S.factor_launches.push_back(Launch{LK_SIDE_FORK, cur_g, 0, 0, 0, 0});
            cur_side = (overlap && pass != 2) ? 1 : 0;
            if (overlap)
                emit_update_launch([&]() {
                    for_fronts([&](i32 s, const FrontDesc &w) {
                        if (ko < w.ns) push_update_region(s, w, kd, ko - kd, ko, std::min(ko + NB_OUT, w.ns), 0, 0, (pass == 2 && !dry && ko - kd <= NB_OUT) ? chain_tile : TILE);
                    });
                });
            if (io < nouter) {
                // narrow blocks (one 64-wide step) and wide ones go to different kernels
                // (and the fronts with <= SMALL_NS pivot columns -- most fronts of the leaf levels -- take one
                // wave per front, four fronts per workgroup, list padded with front = -1)
                for (int cls = 0; cls < 3; ++cls) {              // 0 small, 1 narrow, 2 wide
                    const i64 f_potrf = (i64)S.potrf_tasks.size();
                    for_fronts([&](i32 s, const FrontDesc &w) {
                        if (ko >= w.ns) return;
                        const i32 no = std::min(NB_OUT, w.ns - ko);
                        const int c = (w.ns <= SMALL_NS) ? 0 : (no > NB_IN ? 2 : 1);
                        if (c == cls) S.potrf_tasks.push_back(PotrfTask{s, ko, no, ko});
                    });
                    if (cls == 0) {
                        while (((i64)S.potrf_tasks.size() - f_potrf) % 4) S.potrf_tasks.push_back(PotrfTask{-1, 0, 0, 0});
                        emit(LK_POTRF_SMALL, f_potrf, ((i64)S.potrf_tasks.size() - f_potrf) / 4);
                    } else
                        emit(cls == 2 ? LK_POTRF_WIDE : LK_POTRF, f_potrf, (i64)S.potrf_tasks.size() - f_potrf);
                }
            }
            cur_side = 0;
            {
                // rows below the diagonal block; at the start of a macro column also the other block
                // columns of the macro column (K = [0, kM)); past the last block column of a front,
                // U = -L21 L21' (written)
                emit_update_launch([&]() {
                    for_fronts([&](i32 s, const FrontDesc &w) {
                        const i32 my_nouter = (w.ns + NB_OUT - 1) / NB_OUT;
                        if (io < my_nouter) {
                            push_update_region(s, w, ka, ko - ka, ko, std::min(ko + NB_OUT, w.ns), 0, overlap ? 1 : 2);
                            if (gi == 0 && G > 1)
                                push_update_region(s, w, 0, kM, ko + NB_OUT, std::min(kM + G * NB_OUT, w.ns), 0, 2);
                            // look-ahead: the diagonal block of block column io + 1, K = [k_first(io + 1), ko)
                            if (lookahead && io >= 1 && io + 1 < my_nouter) {
                                const i32 k1 = k_first(io + 1);
                                // (the next block column starts a macro column: the whole block column, see mac_la above)
                                if (ko > k1) push_update_region(s, w, k1, ko - k1, ko + NB_OUT, std::min(ko + 2 * NB_OUT, w.ns), 0, (la_full || mac_first[(size_t)io + 1] == io + 1) ? 2 : 0);
                            }
                        } else if (io == my_nouter) push_update_region(s, w, 0, w.ns, w.ns, w.f, 1, 2);
                    });
                });
            }
            if (overlap && pass != 2) S.factor_launches.push_back(Launch{LK_SIDE_JOIN, cur_g, 0, 0, 0, 0});
            if (io == nouter) break;
            // k_trsm solves the rows below the diagonal block in one pass
            // (thin block columns -- the small fronts of the leaf levels -- take one thread per row instead
            // of 16-row MFMA strips that would be 90 % padding)
            for (int thin = 0; thin < 2; ++thin) {
                const i64 f_trsm = (i64)S.trsm_tasks.size();
                for_fronts([&](i32 s, const FrontDesc &w) {
                    if (ko >= w.ns) return;
                    const i32 no = std::min(NB_OUT, w.ns - ko);
                    if ((no <= TRSM_THIN_W && pass != 2) != (thin == 1)) return;      // (chain items: 64-row strips for every width)
                    const i32 step = thin ? 256 : TRSM_WG_ROWS;
                    // row ranges END on multiples of `step` rows (16-row strips then sit on 128-byte lines of the
                    // line-aligned panel); pad1 = row limit of the task
                    for (i32 r0 = ko + no; r0 < w.f;) {
                        const i32 r1 = std::min(w.f, (r0 / step + 1) * step);
                        S.trsm_tasks.push_back(TrsmTask{s, ko, no, r0, ko, 0, r1, 0});
                        r0 = r1;
                    }
                });
                emit(thin ? LK_TRSM_THIN : LK_TRSM, f_trsm, (i64)S.trsm_tasks.size() - f_trsm);
            }
        }
        };      // block_columns
        // ---- round 6: the dependency-driven form (LK_CHAIN) --------------------------------------------------------------------------------
        // The launches of a block column -- diagonal tiles -> diagonal block -> rows-below tiles -> triangular solve, with their stream forks and joins --
        // are a lock-step over ALL fronts of the level and four or five launch gaps per 256 columns; where a level has few fronts (a pds-class top front,
        // the root front, the blocks of one rank of an 8-GPU job) the chain potrf(io) -> trsm(io) -> diagonal update(io + 1) IS the level's time, and
        // profiles/r05_chain_overlap.txt showed that it never runs beside the rows-below tiles it was forked to hide behind.  Here the SAME tasks (same
        // tiles, same K ranges, same split-K parts: the captured launches of pass 2) become the items of one persistent launch: a workgroup draws an
        // item, waits for the completion counters the item names, runs the task's ordinary device function and publishes its stores with one agent-scope
        // release before it raises its counter (cdna_hip_programming.md, Guideline 16, counter form).  Ticket order = the order of the captured launches
        // = block column major: diagonal tiles, diagonal block, rows-below tiles (+ look-ahead / macro-column tiles), strips of the triangular solve.
        // Every wait names counters raised by EARLIER items only, so no schedule of the workgroups can deadlock (tests/emulate.py asserts it).
        // Adders of one target tile (macro-column tile, look-ahead tile, the block column's own tile or its split-K reduction) are chained through the
        // tile's counter in that order: exactly the order of the launches, so the factor is bit-identical to the launch form (TLPK_CHAIN=0).
        auto build_chain = [&]() {
            if (cap.empty()) return;
            // TLPK_CHAIN_JIT (default 1): the macro-column tiles take their tickets just in time, see `units` below; 0 = in the order of the launches.
            const bool chain_jit = [] { const char *e = std::getenv("TLPK_CHAIN_JIT"); return e && std::atoi(e) != 0; }();
            // TLPK_CHAIN_EARLY (default 1): the strips of a full-width block column do not wait for its diagonal block to be complete -- the diagonal-block role
            // raises the block column's counter on its way (+1 behind each of its first three 64-wide steps, its final signal makes 4) and the strip role waits for
            // the value each of its ten operand blocks needs (kernels.hip: trsm_task_dma).  Here: the strip's item drops the wait (w2), its task names the
            // counter (pad2 = global index + 1), the diagonal block's item is marked (sub = 1).  Same tickets, same data flow, same bits.
            const bool chain_early = !S.shared_device && [] { const char *e = std::getenv("TLPK_CHAIN_EARLY"); return !e || std::atoi(e) != 0; }();
            struct FC { i64 base; i32 nbc, ntr, nsl, stride; };
            std::unordered_map<i32, FC> fc;
            i64 ncnt = 0;
            for (i32 t = t0; t < t1; ++t) {
                const i32 s = S.level_fronts[t];
                if (!in_scope(s) || !chain_front[(size_t)s]) continue;
                const FrontDesc &w = S.fronts[s];
                FC c; c.base = ncnt; c.nbc = (w.ns + NB_OUT - 1) / NB_OUT; c.ntr = (w.f + TILE - 1) / TILE; c.nsl = (w.f + 63) / 64;
                c.stride = 5 + 2 * c.ntr + c.nsl;
                ncnt += (i64)c.nbc * c.stride;
                fc[s] = c;
            }
            const i64 ticket_idx = S.chain_counters, cbase = S.chain_counters + 1;      // global index of local counter q: cbase + q
            std::vector<i32> expect((size_t)ncnt, 0);                                   // signals handed out so far, per local counter
            auto new_counter = [&]() { expect.push_back(0); return (i64)expect.size() - 1; };
            auto c_dg = [&](const FC &c, i32 io, i32 pos) { return c.base + (i64)io * c.stride + pos; };                  // diagonal block of io: tiles (ko, ko) | (ko + 128, ko) | (ko + 128, ko + 128)
            auto c_pf = [&](const FC &c, i32 io) { return c.base + (i64)io * c.stride + 3; };                             // the diagonal block is factored
            auto c_dq = [&](const FC &c, i32 io) { return c.base + (i64)io * c.stride + 4; };                             // the 64 x 64 tiles of the diagonal block's last (short) update
            auto c_tg = [&](const FC &c, i32 io, i32 tr, i32 cj) { return c.base + (i64)io * c.stride + 5 + 2 * tr + cj; };     // target tile (rows 128 tr .., column tile cj of block column io)
            auto c_ts = [&](const FC &c, i32 io, i32 sl) { return c.base + (i64)io * c.stride + 5 + 2 * c.ntr + sl; };    // rows [64 sl, 64 sl + 64) are solved in block column io
            std::unordered_map<i64, char> covered;                // target tiles of a diagonal block that a 64 x 64 tile waits for (the diagonal block then needs not)
            const i64 first_item = (i64)S.chain_items.size();
            bool bad = false;
            auto G = [&](i64 q) { return (i32)(cbase + q); };
            // the counter an adder of target tile (i0, j0) raises, or -1 (targets in the update matrix: read by the next level's extend-add launch)
            auto target_counter = [&](const FC &c, const FrontDesc &w, i32 i0, i32 j0) -> i64 {
                if (j0 >= w.ns) return -1;
                const i32 io = j0 / NB_OUT, ko = io * NB_OUT;
                if (i0 < ko + NB_OUT) return c_dg(c, io, (i0 == ko) ? 0 : 1 + (j0 - ko) / TILE);
                return c_tg(c, io, i0 / TILE, (j0 - ko) / TILE);
            };
            // operand rows [r0, r0 + 128) of an update tile whose K range ends in block column io_k: hand-over flags of the strips that solved them
            auto operand_wait = [&](const FC &c, const FrontDesc &w, i32 io_k, i32 r0, i32 &wq, i32 &nq) {
                const i32 s0 = r0 / 64, s1 = (std::min(r0 + TILE, w.f) - 1) / 64;
                for (i32 sl = s0; sl <= s1; ++sl) if (expect[(size_t)c_ts(c, io_k, sl)] != 1) bad = true;      // no strip (or two) for these rows: a bug
                wq = G(c_ts(c, io_k, s0)); nq = s1 - s0 + 1;
            };
            // Tickets are priorities, and a workgroup keeps the item it drew: at the start of a macro column the launch order puts ~900 tiles with K = [0, kM) --
            // the long update of the macro column's OTHER block columns, hundreds of microseconds each even cut by K length -- in front of the chain's next links,
            // which then wait for a free workgroup (profiles/r06_chain_trace_pds*.txt: 0.3 - 0.7 ms stalls at every macro column start; putting them all behind
            // the strips of that block column, round 6's first try, only moved the stall to the next block column).  Just in time: the tiles that feed block
            // column io_t are held back until block column io_t - 2 -- they take their tickets behind its rows-below tiles and in front of its strips, two links
            // of the chain before they are needed, one block column's worth at a time.  A held tile travels with its split-K parts and its reduction.  Every
            // adder of a target tile is still created before the later adders of that tile (the look-ahead tiles of block column io_t come with io_t - 1, its
            // own tiles with io_t): same order of the sums, same bits as the launch form.
            struct Unit { i64 first, count, red; };                    // update tasks [first, first + count) (one tile: itself, or its split-K parts) and its reduce task (or -1)
            std::map<i32, std::vector<Unit>> held;                       // target block column -> units
            std::unordered_map<i32, i64> slot_red;                       // split-K scratch slot -> the counter of the reduction that owns it (slots are unique inside a chain launch)
            std::unordered_map<i64, i64> red_rc;                         // reduce task -> its counter
            i32 io_cur = -1;                                             // block column of the last diagonal block seen
            auto update_item = [&](i64 q) {
                        const UpdateTask &u = S.update_tasks[(size_t)q];
                        const FrontDesc &w = S.fronts[u.front];
                        const FC &c = fc.at(u.front);
                        ChainItem it{CR_UPDATE, (i32)q, 0, 0, 0, 0, 0, 0, 0, -1, 0, -1};
                        const i32 io_k = (u.k0 + u.kw - 1) / NB_OUT;
                        operand_wait(c, w, io_k, u.i0, it.w0, it.n0); it.need0 = 1;
                        if (u.j0 != u.i0) { operand_wait(c, w, io_k, u.j0, it.w1, it.n1); it.need1 = 1; }
                        if (u.pad1) {
                            const auto f = slot_red.find(u.pad1 - 1);
                            if (f == slot_red.end()) { bad = true; return; }
                            it.sig = G(f->second); ++expect[(size_t)f->second];
                        } else if (u.pad2) {
                            // a 64 x 64 tile of the diagonal block: ordered behind the earlier adders of the 128 x 128 target tile that holds it (look-ahead / macro-column
                            // tiles) through that tile's counter, which it does NOT raise -- its siblings must not wait for it --; all of them raise one counter of their own
                            const i32 io = u.j0 / NB_OUT, ko = io * NB_OUT;
                            const i64 tc = target_counter(c, w, ko + ((u.i0 - ko) & ~(TILE - 1)), ko + ((u.j0 - ko) & ~(TILE - 1)));
                            if (u.j0 >= w.ns || tc < 0 || u.i0 >= ko + NB_OUT) { bad = true; return; }
                            if (expect[(size_t)tc] > 0) { it.w2 = G(tc); it.need2 = expect[(size_t)tc]; }
                            covered[tc] = 1;
                            it.sig = G(c_dq(c, io)); ++expect[(size_t)c_dq(c, io)];
                        } else {
                            const i64 tc = target_counter(c, w, u.i0, u.j0);
                            if (tc >= 0) {
                                if (expect[(size_t)tc] > 0) { it.w2 = G(tc); it.need2 = expect[(size_t)tc]; }     // the earlier adder(s) of this tile
                                it.sig = G(tc); ++expect[(size_t)tc];
                            }
                        }
                        S.chain_items.push_back(it);
            };
            auto reduce_items = [&](i64 q) {
                const UpdateTask &r = S.reduce_tasks[(size_t)q];
                const FrontDesc &w = S.fronts[r.front];
                const FC &c = fc.at(r.front);
                const i64 rc = red_rc.at(q);
                if (expect[(size_t)rc] != r.kw) { bad = true; return; }
                const i64 tc = target_counter(c, w, r.i0, r.j0);
                for (i32 sub = 0; sub < RED_SPLIT; ++sub) {
                    ChainItem it{CR_REDUCE, (i32)q, sub, G(rc), 1, r.kw, 0, 0, 0, -1, 0, -1};
                    if (tc >= 0) {
                        if (expect[(size_t)tc] > 0) { it.w2 = G(tc); it.need2 = expect[(size_t)tc]; }
                        it.sig = G(tc);
                    }
                    S.chain_items.push_back(it);
                }
                if (tc >= 0) expect[(size_t)tc] += RED_SPLIT;
            };
            auto emit_units = [&](const std::vector<Unit> &us) {           // the tiles first, then their reductions (the order of a launch pair)
                for (const Unit &u : us) for (i64 q = u.first; q < u.first + u.count && !bad; ++q) update_item(q);
                for (const Unit &u : us) if (u.red >= 0 && !bad) reduce_items(u.red);
            };
            auto flush_held = [&](i32 io_upto) {                           // the held tiles of the block columns <= io_upto, in block-column order
                while (!held.empty() && held.begin()->first <= io_upto && !bad) { emit_units(held.begin()->second); held.erase(held.begin()); }
            };
            for (size_t ci = 0; ci < cap.size() && !bad; ++ci) {
                const Cap &L = cap[ci];
                if (L.kind == LK_UPDATE) {
                    // split-K parts of this launch: scratch slot -> the counter of the reduction that owns it
                    std::unordered_map<i32, i64> slot_task;                // slot -> reduce task
                    const bool has_red = ci + 1 < cap.size() && cap[ci + 1].kind == LK_UPDATE_REDUCE;
                    if (has_red) {
                        const Cap &R = cap[ci + 1];
                        for (i64 q = R.first; q < R.first + R.count; ++q) {
                            const UpdateTask &r = S.reduce_tasks[(size_t)q];
                            const i64 rc = new_counter();
                            red_rc[q] = rc;
                            for (i32 sp = 0; sp < r.kw; ++sp) { slot_red[r.k0 + sp] = rc; slot_task[r.k0 + sp] = q; }
                        }
                    }
                    std::vector<Unit> now;
                    i64 nred_seen = 0;
                    for (i64 q = L.first; q < L.first + L.count && !bad;) {
                        const UpdateTask &u = S.update_tasks[(size_t)q];
                        Unit un{q, 1, -1};
                        if (u.pad1) {                                    // the consecutive parts of one tile
                            const auto f = slot_task.find(u.pad1 - 1);
                            if (f == slot_task.end()) { bad = true; break; }
                            un.red = f->second; ++nred_seen;
                            while (q + un.count < L.first + L.count) {
                                const UpdateTask &v = S.update_tasks[(size_t)(q + un.count)];
                                const auto g = v.pad1 ? slot_task.find(v.pad1 - 1) : slot_task.end();
                                if (g == slot_task.end() || g->second != un.red) break;
                                ++un.count;
                            }
                        }
                        const i32 io_t = (u.j0 < S.fronts[u.front].ns) ? u.j0 / NB_OUT : -1;
                        if (chain_jit && io_cur >= 0 && io_t >= io_cur + 3) held[io_t].push_back(un); else now.push_back(un);
                        q += un.count;
                    }
                    if (has_red && nred_seen != cap[ci + 1].count) bad = true;      // every reduction belongs to exactly one tile of this launch
                    emit_units(now);
                    if (has_red) ++ci;                                   // the reduce launch is consumed
                } else if (L.kind == LK_POTRF || L.kind == LK_POTRF_WIDE) {
                    for (i64 q = L.first; q < L.first + L.count; ++q) {
                        const PotrfTask &pt = S.potrf_tasks[(size_t)q];
                        io_cur = std::max(io_cur, pt.k0 / NB_OUT);
                        const FC &c = fc.at(pt.front);
                        const i32 io = pt.k0 / NB_OUT;
                        ChainItem it{CR_POTRF, (i32)q, (chain_early && pt.nb == NB_OUT) ? 1 : 0, 0, 0, 0, 0, 0, 0, -1, 0, G(c_pf(c, io))};
                        // waits: the counter of the block's 64 x 64 tiles (they waited for the adders of their target tiles themselves), and every target tile
                        // with adders that no 64 x 64 tile stands behind -- at most three counters in all
                        i64 wl[4]; int nw = 0;
                        if (expect[(size_t)c_dq(c, io)] > 0) wl[nw++] = c_dq(c, io);
                        for (int pos = 0; pos < 3; ++pos) { const i64 d = c_dg(c, io, pos); if (expect[(size_t)d] > 0 && !covered.count(d)) wl[nw++] = d; }
                        if (nw > 3) { bad = true; break; }
                        if (nw > 0) { it.w0 = G(wl[0]); it.n0 = 1; it.need0 = expect[(size_t)wl[0]]; }
                        if (nw > 1) { it.w1 = G(wl[1]); it.n1 = 1; it.need1 = expect[(size_t)wl[1]]; }
                        if (nw > 2) { it.w2 = G(wl[2]); it.need2 = expect[(size_t)wl[2]]; }
                        ++expect[(size_t)c_pf(c, io)];
                        S.chain_items.push_back(it);
                    }
                } else if (L.kind == LK_TRSM) {
                    flush_held(io_cur + 2);                              // just in time: behind this block column's rows-below tiles, in front of its strips
                    for (i64 q = L.first; q < L.first + L.count; ++q) {
                        const TrsmTask &tt = S.trsm_tasks[(size_t)q];
                        const FC &c = fc.at(tt.front);
                        const i32 io = tt.k0 / NB_OUT, tr = tt.row0 / TILE;
                        if (expect[(size_t)c_pf(c, io)] != 1 || tt.pad1 > (tt.row0 / 64 + 1) * 64) { bad = true; break; }
                        ChainItem it{CR_TRSM, (i32)q, 0, 0, 0, 0, 0, 0, 0, G(c_pf(c, io)), 1, G(c_ts(c, io, tt.row0 / 64))};
                        if (tr * TILE >= tt.k0 + NB_OUT) {              // (a strip inside the diagonal tiles' rows -- a narrow last block column -- is released by the diagonal block alone)
                            const i64 g0 = c_tg(c, io, tr, 0), g1 = c_tg(c, io, tr, 1);
                            if (expect[(size_t)g0] > 0) { it.w0 = G(g0); it.n0 = 1; it.need0 = expect[(size_t)g0]; }
                            if (expect[(size_t)g1] > 0) { it.w1 = G(g1); it.n1 = 1; it.need1 = expect[(size_t)g1]; }
                        }
                        if (chain_early && tt.nb == NB_OUT) { it.w2 = -1; it.need2 = 0; S.trsm_tasks[(size_t)q].pad2 = G(c_pf(c, io)) + 1; }
                        ++expect[(size_t)c_ts(c, io, tt.row0 / 64)];
                        S.chain_items.push_back(it);
                    }
                } else bad = true;                                       // (no other kind is ever captured)
            }
            flush_held(INT32_MAX);
            // the values the diagonal blocks and the strips wait for must be FINAL: nothing after them may add to their tiles
            for (i64 q = first_item; q < (i64)S.chain_items.size() && !bad; ++q) {
                const ChainItem &it = S.chain_items[(size_t)q];
                if (it.role != CR_POTRF && it.role != CR_TRSM) continue;
                if (it.n0 && expect[(size_t)(it.w0 - cbase)] != it.need0) bad = true;
                if (it.n1 && expect[(size_t)(it.w1 - cbase)] != it.need1) bad = true;
                if (it.role == CR_POTRF && it.w2 >= 0 && expect[(size_t)(it.w2 - cbase)] != it.need2) bad = true;
            }
            if (bad) { S.error = "internal: inconsistent chain schedule"; return; }
            S.factor_launches.push_back(Launch{LK_CHAIN, cur_g, first_item, (i64)S.chain_items.size() - first_item, 0, (i32)ticket_idx});
            S.chain_counters += 1 + (i64)expect.size();
        };
        {
            // TLPK_CHAIN: 0 = off, 1 = every level that has a front with more than one block column, unset = auto: the levels the look-ahead rule above names
            // (at most TLPK_CHAIN_MAX_FRONTS = 8 of this rank's fronts have more than one block column, none more than 12 288 pivot columns).  The chain's
            // diagonal-block role is the round-5 DPP kernel: the older block kernels (TLPK_POTRF_MODE != 3, diagnostics) keep the launches.
            const int chain_env = [] { const char *e = std::getenv("TLPK_CHAIN"); return e ? std::atoi(e) : -1; }();
            const i32 chain_max = [] { const char *e = std::getenv("TLPK_CHAIN_MAX_FRONTS"); return e ? std::max(1, std::atoi(e)) : 8; }();
            const bool potrf_default = [] {
                const char *m = std::getenv("TLPK_POTRF_MODE");
                return (!m || (std::atoi(m) & 3) == 3) && !std::getenv("TLPK_POTRF_WAVE") && !std::getenv("TLPK_POTRF_PAIR") && !std::getenv("TLPK_POTRF_DYN");
            }();
            i32 nbig = 0, ns_big = 0;
            for (i32 t = t0; t < t1; ++t) {
                const i32 sf = S.level_fronts[t];
                if (!S.front_local[sf] || S.front_single[sf]) continue;
                if (S.fronts[sf].ns > NB_OUT) { ++nbig; ns_big = std::max(ns_big, S.fronts[sf].ns); }
            }
            // ... and the widest of them has at least TLPK_CHAIN_MIN_NS pivot columns: a front of two or three block columns has too few items for the hand-overs
            // (a few microseconds each) to beat the launches it replaces (25fv47-class LPs; the lower levels of a pds-class LP)
            const i32 chain_min_ns = [] { const char *e = std::getenv("TLPK_CHAIN_MIN_NS"); return e ? std::atoi(e) : 769; }();
            const bool use_chain = chain_env != 0 && potrf_default && UPD_SLOTS == 0 && nbig >= 1 &&
                                   (chain_env > 0 || (nbig <= chain_max && ns_big <= 12288 && ns_big >= chain_min_ns));
            bool any = false;
            if (use_chain)
                for (i32 t = t0; t < t1; ++t) {
                    const i32 sf = S.level_fronts[t];
                    if (S.front_local[sf] && !S.front_single[sf] && S.fronts[sf].ns > NB_OUT) { chain_front[(size_t)sf] = 1; any = any || in_scope(sf); }
                    if (chain_front[(size_t)sf] && in_scope(sf)) {       // the algorithmic update flops of its columns (the formula of step 12) now run inside k_chain
                        const FrontDesc &w = S.fronts[sf];
                        for (i32 c = 0; c < w.ns; ++c) {
                            const double l = (double)S.colcount[w.col0 + c] - (double)(std::min((c / NB_OUT + 1) * NB_OUT, w.ns) - c);
                            if (l > 0) S.flops_update_alg_chain += l * l;
                        }
                    }
                }
            if (!any) { pass = 0; block_columns(); }
            else {
                pass = 1; block_columns();
                pass = 2; cap.clear(); chain_slot_base = 0; block_columns();
                build_chain();
                pass = 0;
            }
        }
        // (c) extend-add, U part (every U of this level has been written by now)
        push_ea(true);
    };
    for (cur_g = 0; cur_g < S.ngroups; ++cur_g)
        for (i32 d = S.nlevels - 1; d >= 1; --d) factor_level(d);
    cur_g = -1;
    if (S.nlevels > 0) factor_level(0);
    // split-K scratch: every stream (group x side) gets its own region, launches of one stream reuse it;
    // make the slot numbers absolute
    {
        std::vector<i64> base(region_slots.size() + 1, 0);
        for (size_t r = 0; r < region_slots.size(); ++r) base[r + 1] = base[r] + region_slots[r];
        S.spart_len = base.back() * (i64)TILE * TILE;
        for (const Launch &L : S.factor_launches) {
            if (L.kind != LK_UPDATE && L.kind != LK_UPDATE_REDUCE) continue;
            const size_t region = (size_t)((L.group + 1) * 2 + L.side);
            if (region >= region_slots.size() || base[region] == 0) continue;
            for (i64 q = L.first; q < L.first + L.count; ++q) {
                if (L.kind == LK_UPDATE) { if (S.update_tasks[q].pad1) S.update_tasks[q].pad1 += (i32)base[region]; }
                else S.reduce_tasks[q].k0 += (i32)base[region];
            }
        }
        for (const Launch &L : S.factor_launches) {              // the split-K tasks inside the chain launches (region of the group's main stream)
            if (L.kind != LK_CHAIN) continue;
            const size_t region = (size_t)((L.group + 1) * 2);
            if (region >= region_slots.size() || base[region] == 0) continue;
            for (i64 q = L.first; q < L.first + L.count; ++q) {
                const ChainItem &it = S.chain_items[(size_t)q];
                if (it.role == CR_UPDATE) { if (S.update_tasks[(size_t)it.task].pad1) S.update_tasks[(size_t)it.task].pad1 += (i32)base[region]; }
                else if (it.role == CR_REDUCE && it.sub == 0) S.reduce_tasks[(size_t)it.task].k0 += (i32)base[region];
            }
        }
    }
    spt.mark("schedule: fwd");
    // ---------------- forward solve: deepest level first ----------------
    // (thin but TALL fronts keep the workgroup-per-row-chunk kernels: one wave walking 1000 rows is slower)
    auto is_small = [&](i32 s) { return S.fronts[s].ns <= SMALL_NS && S.fronts[s].f - S.fronts[s].ns <= SMALL_ROWS && s != S.root_front; };
    // Persistent sweeps (default): the block steps of a level's triangular solves run inside ONE launch per
    // direction; a solved SOLVE_NB-wide block is handed to the workgroups that need it through a flag word
    // per (front, block).  TLPK_SWEEP=0 keeps one launch per block step (the round-1 schedule).
    S.sweep = true;
    if (const char *e = std::getenv("TLPK_SWEEP")) S.sweep = std::atoi(e) != 0;
    S.n_sweep_flags = 0;
    for (size_t s = 0; s < S.fronts.size(); ++s) {
        FrontDesc &w = S.fronts[s];
        w.flagoff = -1;
        if (!S.front_local[s] || S.front_single[s] || is_small((i32)s)) continue;
        w.flagoff = 0;                       // handled by the sweep kernels (hand-over words are indexed by column)
        S.n_sweep_flags += 1;
    }
    // (TLPK_SOLVE_SIDE=1, experiment, OFF: measured neutral on C4 / north-star -- 51.6 vs 51.5, 136.3 vs 136.5 ms -- and SLOWER on the latency-bound LPs, 25fv47 class 1.17
    // vs 0.99 ms, pds class 14.05 vs 13.83: a fork / join costs more than the launch it takes off the chain; profiles/r06_solve_side.txt)
    const i64 solve_merge = [] { const char *e = std::getenv("TLPK_SOLVE_MERGE"); return e ? (i64)std::max(0, std::atoi(e)) : (i64)256; }();
    const bool solve_side = [] { const char *e = std::getenv("TLPK_SOLVE_SIDE"); return e && std::atoi(e) != 0; }();
    auto fwd_level = [&](i32 d) {
        const i32 t0 = S.level_ptr[d], t1 = S.level_ptr[d + 1];
        const bool root_level = (d == 0 && S.root_front >= 0);
        i64 small_first = 0, small_count = 0;
        {
            const i64 first = (i64)S.fwd_gather_tasks.size();
            for (i32 t = t0; t < t1; ++t) {
                const i32 s = S.level_fronts[t];
                if (!in_scope(s)) continue;
                const FrontDesc &w = S.fronts[s];
                if (w.nchild == 0 && w.f == w.ns) continue;
                // leaves only clear their contribution vector (rows >= ns)
                // nb = rows of the task: 256 (one thread per row), or 32 = 8 lanes per row for a front whose rows collect many
                // entries each (the root front of a block-angular LP: one per diagonal block)
                const double per_row = (double)(S.gth_ptr[(size_t)w.rowoff + w.f] - S.gth_ptr[(size_t)w.rowoff]) / std::max(1, w.f);
                const i32 step = (per_row >= GATHER_WIDE_PER_ROW) ? SOLVE_ROWS / 8 : SOLVE_ROWS;
                for (i32 r0 = (w.nchild == 0) ? (w.ns / SOLVE_ROWS) * SOLVE_ROWS : 0; r0 < w.f; r0 += step)
                    S.fwd_gather_tasks.push_back(SolveTask{s, 0, step, r0, 0, 0, 0, 0});
            }
            push_launch(S.fwd_launches, LK_FWD_GATHER, first, (i64)S.fwd_gather_tasks.size() - first);
        }
        if (root_level) S.fwd_launches.push_back(Launch{LK_ALLREDUCE_ROOT, -1, 0, 0});
        // small fronts (<= SMALL_NS pivot columns: most fronts of the leaf levels): diagonal solve and
        // update of the rows below by ONE wave per front, four fronts per workgroup (a 256-thread
        // workgroup per front and kernel is mostly fixed latency); padded to a multiple of 4
        {
            const i64 first = (i64)S.fwd_small_tasks.size();
            for (i32 t = t0; t < t1; ++t) {
                const i32 s = S.level_fronts[t];
                if (in_scope(s) && is_small(s)) S.fwd_small_tasks.push_back(SolveTask{s, 0, S.fronts[s].ns, 0, 0, 0, 0, 0});
            }
            while (((i64)S.fwd_small_tasks.size() - first) % 4) S.fwd_small_tasks.push_back(SolveTask{-1, 0, 0, 0, 0, 0, 0, 0});
            // Round 6: the small fronts of a level beside its sweep (the group's side stream, forked behind the gather and joined in front of the next level's gather)
            // when the level has both: different fronts of one level, nothing in common but the gathered right-hand side.  One launch off the level's chain -- what a
            // latency-bound LP pays per launch, and on the north-star LP the small-front kernels were 0.8 of the 6.5 ms of a solve.  MEASURED and left OFF (see `solve_side`).
            small_first = first; small_count = ((i64)S.fwd_small_tasks.size() - first) / 4;
        }
        i32 max_ns = 0;
        for (i32 t = t0; t < t1; ++t) if (in_scope(S.level_fronts[t]) && !is_small(S.level_fronts[t])) max_ns = std::max(max_ns, S.fronts[S.level_fronts[t]].ns);
        if (S.sweep) {
            // Items in hand-out order (workgroups draw them from a ticket counter): chunk index first, front
            // second, so that an item only ever waits for items with a smaller ticket -- those are held by
            // workgroups that are already running, whatever the dispatch order (no deadlock), and the fronts of
            // the level advance side by side.
            const i64 first = (i64)S.fwd_sweep_tasks.size();
            i32 max_chunks = 0;
            for (i32 t = t0; t < t1; ++t) {
                const i32 s = S.level_fronts[t];
                if (!in_scope(s) || is_small(s)) continue;
                const FrontDesc &w = S.fronts[s];
                max_chunks = std::max(max_chunks, (w.ns + SWEEP_NB - 1) / SWEEP_NB + (w.f > w.ns ? (w.f + SOLVE_NB - 1) / SOLVE_NB - w.ns / SOLVE_NB : 0));
            }
            for (i32 ci = 0; ci < max_chunks; ++ci)
                for (i32 t = t0; t < t1; ++t) {
                    const i32 s = S.level_fronts[t];
                    if (!in_scope(s) || is_small(s)) continue;
                    const FrontDesc &w = S.fronts[s];
                    const i32 nblk = (w.ns + SWEEP_NB - 1) / SWEEP_NB;
                    if (ci < nblk) S.fwd_sweep_tasks.push_back(SolveTask{s, ci * SWEEP_NB, std::min(SWEEP_NB, w.ns - ci * SWEEP_NB), 0, 1, ci, 0, 0});
                    else {
                        // rows below the pivot block in chunks that END on multiples of SOLVE_NB rows (line-aligned loads;
                        // only the first chunk of a front is ragged)
                        const i32 q = ci - nblk;
                        const i32 r0 = (q == 0) ? w.ns : (w.ns / SOLVE_NB + q) * SOLVE_NB;
                        const i32 r1 = std::min(w.f, (w.ns / SOLVE_NB + q + 1) * SOLVE_NB);
                        if (r0 < w.f) S.fwd_sweep_tasks.push_back(SolveTask{s, r0, r1 - r0, 0, 0, nblk, 0, 0});
                    }
                }
            i64 sweep_count = (i64)S.fwd_sweep_tasks.size() - first;
            // Round 6: on a level whose sweep is small (at most TLPK_SOLVE_MERGE = 256 items) the small fronts ride in the sweep's launch, as items of their own behind
            // the sweep's (slot = 2, k0 = a group of four small-front tasks; no dependencies: any ticket will do) -- one launch per level and direction less where a
            // launch costs more than the fronts in it.  Same bodies, same arithmetic.
            bool merged = false;
            if (solve_merge > 0 && small_count > 0 && sweep_count > 0 && sweep_count <= solve_merge) {
                for (i64 g = 0; g < small_count; ++g) {
                    i32 fr = -1;
                    for (int u = 0; u < 4; ++u) if (S.fwd_small_tasks[(size_t)(small_first + 4 * g + u)].front >= 0) { fr = S.fwd_small_tasks[(size_t)(small_first + 4 * g + u)].front; break; }
                    S.fwd_sweep_tasks.push_back(SolveTask{fr, (i32)(small_first / 4 + g), 0, 0, 2, 0, 0, 0});
                }
                sweep_count += small_count; small_count = 0; merged = true;
            }
            const bool beside = solve_side && small_count > 0 && sweep_count > 0;
            if (beside) { S.fwd_launches.push_back(Launch{LK_SIDE_FORK, cur_g, 0, 0, 0, 0}); cur_side = 1; }
            push_launch(S.fwd_launches, LK_FWD_SMALL, small_first, small_count);
            cur_side = 0;
            push_launch(S.fwd_launches, LK_FWD_SWEEP, first, sweep_count);
            if (merged) S.fwd_launches.back().pad = 1;
            if (beside) S.fwd_launches.push_back(Launch{LK_SIDE_JOIN, cur_g, 0, 0, 0, 0});
            small_count = 0;
            max_ns = 0;        // no per-block launches
        }
        push_launch(S.fwd_launches, LK_FWD_SMALL, small_first, small_count);      // (TLPK_SWEEP=0: in stream order)
        for (i32 kb = 0; kb < max_ns; kb += SOLVE_NB) {
            const i64 f_diag = (i64)S.fwd_diag_tasks.size(), f_upd = (i64)S.fwd_update_tasks.size();
            // pass 0: the look-ahead workgroups (first row chunk: they also solve the next diagonal
            // block) of every front, so that they start with the launch; pass 1: the other chunks
            for (int pass = 0; pass < 2; ++pass)
            for (i32 t = t0; t < t1; ++t) {
                const i32 s = S.level_fronts[t];
                if (!in_scope(s) || is_small(s)) continue;
                const FrontDesc &w = S.fronts[s];
                if (kb >= w.ns) continue;
                const i32 nb = std::min(SOLVE_NB, w.ns - kb);
                const i32 next_nb = std::min(SOLVE_NB, w.ns - (kb + nb));      // <= 0: last block
                if (pass == 0 && kb == 0) S.fwd_diag_tasks.push_back(SolveTask{s, kb, nb, 0, 0, 0, 0, 0});
                for (i32 r0 = kb + nb; r0 < w.f; r0 += SOLVE_ROWS) {
                    const bool first = (r0 == kb + nb);
                    if ((pass == 0) != first) continue;
                    S.fwd_update_tasks.push_back(SolveTask{s, kb, nb, r0, 0, (first && next_nb > 0) ? next_nb : 0, 0, 0});
                }
            }
            push_launch(S.fwd_launches, LK_FWD_DIAG, f_diag, (i64)S.fwd_diag_tasks.size() - f_diag);
            push_launch(S.fwd_launches, LK_FWD_UPDATE, f_upd, (i64)S.fwd_update_tasks.size() - f_upd);
        }
    };
    // One solve schedule for all stream groups (round 5, default; TLPK_SOLVE_ONE_GROUP=0 restores one schedule per group).  The stream groups exist
    // for the factorisation, whose launches leave tails that a second group fills.  The solve's big launches are the persistent, ticketed sweeps:
    // one of them fills the chip, and a second group's leaf-level launches then crawl beside the first group's sweep -- in the paired solve the two
    // groups' forward sweeps ran one after the other (629 + 650 us, profiles/r04_solve_timeline.txt).  With scope -1 a level's launch holds the
    // fronts of every group and runs on the main stream; per-item arithmetic is unchanged (bit-identical results).
    const bool solve_one_group = [&] { const char *e = std::getenv("TLPK_SOLVE_ONE_GROUP"); return S.ngroups >= 2 && (!e || std::atoi(e) != 0); }();
    S.solve_single_stream = solve_one_group || S.ngroups <= 1;
    cur_g = solve_one_group ? -1 : 0;
    for (; cur_g < (solve_one_group ? 0 : S.ngroups); ++cur_g)
        for (i32 d = S.nlevels - 1; d >= 1; --d) fwd_level(d);
    cur_g = -1;
    if (S.nlevels > 0) fwd_level(0);
    spt.mark("schedule: bwd");
    // ---------------- backward solve: root level first ----------------
    // Column-oriented: launch 0 of a level removes the rows below the pivot block (known from the
    // ancestors) from every column block of every front and solves each front's last block; launch
    // b >= 1 removes the block solved by launch b-1 from the column blocks before it and solves the
    // next one.  SolveTask fields here: k0/nb = target column block, row0/slot = first source row
    // and number of source rows, nslot != 0 = also solve the diagonal block k0.
    auto bwd_level = [&](i32 d) {
        const i32 t0 = S.level_ptr[d], t1 = S.level_ptr[d + 1];
        i64 small_first = 0, small_count = 0;
        {
            const i64 first = (i64)S.bwd_small_tasks.size();
            for (i32 t = t0; t < t1; ++t) {
                const i32 s = S.level_fronts[t];
                if (in_scope(s) && is_small(s)) S.bwd_small_tasks.push_back(SolveTask{s, 0, S.fronts[s].ns, 0, 0, 0, 0, 0});
            }
            while (((i64)S.bwd_small_tasks.size() - first) % 4) S.bwd_small_tasks.push_back(SolveTask{-1, 0, 0, 0, 0, 0, 0, 0});
            small_first = first; small_count = ((i64)S.bwd_small_tasks.size() - first) / 4;      // (beside the level's sweep: see fwd_level)
        }
        i32 max_ns = 0;
        for (i32 t = t0; t < t1; ++t) if (in_scope(S.level_fronts[t]) && !is_small(S.level_fronts[t])) max_ns = std::max(max_ns, S.fronts[S.level_fronts[t]].ns);
        i32 nblk = (max_ns + SOLVE_NB - 1) / SOLVE_NB;
        if (S.sweep) {
            // hand-out order: distance of the column block from the END of its front first (a block waits for the
            // later blocks of its own front only), front second
            const i64 first = (i64)S.bwd_sweep_tasks.size();
            const i32 nblk64 = (max_ns + SWEEP_NB - 1) / SWEEP_NB;
            for (i32 dd = 0; dd < nblk64; ++dd)
                for (i32 t = t0; t < t1; ++t) {
                    const i32 s = S.level_fronts[t];
                    if (!in_scope(s) || is_small(s)) continue;
                    const FrontDesc &w = S.fronts[s];
                    const i32 my_nblk = (w.ns + SWEEP_NB - 1) / SWEEP_NB;
                    if (dd >= my_nblk) continue;
                    const i32 kb = my_nblk - 1 - dd;
                    S.bwd_sweep_tasks.push_back(SolveTask{s, kb * SWEEP_NB, std::min(SWEEP_NB, w.ns - kb * SWEEP_NB), w.ns, w.f - w.ns, dd, 0, 0});
                }
            i64 sweep_count = (i64)S.bwd_sweep_tasks.size() - first;
            bool merged = false;
            if (solve_merge > 0 && small_count > 0 && sweep_count > 0 && sweep_count <= solve_merge) {      // (see fwd_level; here nslot = -2 marks the group)
                for (i64 g = 0; g < small_count; ++g) {
                    i32 fr = -1;
                    for (int u = 0; u < 4; ++u) if (S.bwd_small_tasks[(size_t)(small_first + 4 * g + u)].front >= 0) { fr = S.bwd_small_tasks[(size_t)(small_first + 4 * g + u)].front; break; }
                    S.bwd_sweep_tasks.push_back(SolveTask{fr, (i32)(small_first / 4 + g), 0, 0, 0, -2, 0, 0});
                }
                sweep_count += small_count; small_count = 0; merged = true;
            }
            const bool beside = solve_side && small_count > 0 && sweep_count > 0;
            if (beside) { S.bwd_launches.push_back(Launch{LK_SIDE_FORK, cur_g, 0, 0, 0, 0}); cur_side = 1; }
            push_launch(S.bwd_launches, LK_BWD_SMALL, small_first, small_count);
            cur_side = 0;
            push_launch(S.bwd_launches, LK_BWD_SWEEP, first, sweep_count);
            if (merged) S.bwd_launches.back().pad = 1;
            if (beside) S.bwd_launches.push_back(Launch{LK_SIDE_JOIN, cur_g, 0, 0, 0, 0});
            small_count = 0;
            nblk = 0;
        }
        push_launch(S.bwd_launches, LK_BWD_SMALL, small_first, small_count);      // (TLPK_SWEEP=0: in stream order)
        for (i32 b = 0; b < nblk; ++b) {
            const i64 f_upd = (i64)S.bwd_update_tasks.size();
            // pass 0: the workgroups that also solve a diagonal block (critical path) start first
            for (int pass = 0; pass < 2; ++pass)
            for (i32 t = t0; t < t1; ++t) {
                const i32 s = S.level_fronts[t];
                if (!in_scope(s) || is_small(s)) continue;
                const FrontDesc &w = S.fronts[s];
                const i32 my_nblk = (w.ns + SOLVE_NB - 1) / SOLVE_NB;
                if (b >= my_nblk) continue;
                const i32 ksrc = my_nblk - b;                       // source block (== my_nblk: rows below the pivot block)
                const i32 row0 = (b == 0) ? w.ns : ksrc * SOLVE_NB;
                const i32 nrows = (b == 0) ? (w.f - w.ns) : std::min(SOLVE_NB, w.ns - row0);
                for (i32 J = ksrc - 1; J >= 0; --J) {
                    const bool diag = (J == ksrc - 1);
                    if ((pass == 0) != diag) continue;
                    if (!diag && nrows == 0) continue;
                    S.bwd_update_tasks.push_back(SolveTask{s, J * SOLVE_NB, std::min(SOLVE_NB, w.ns - J * SOLVE_NB), row0, nrows, diag ? 1 : 0, 0, 0});
                }
            }
            push_launch(S.bwd_launches, LK_BWD_UPDATE, f_upd, (i64)S.bwd_update_tasks.size() - f_upd);
        }
    };
    cur_g = -1;
    if (S.nlevels > 0) bwd_level(0);
    cur_g = solve_one_group ? -1 : 0;
    for (; cur_g < (solve_one_group ? 0 : S.ngroups); ++cur_g)
        for (i32 d = 1; d < S.nlevels; ++d) bwd_level(d);
    spt.mark(nullptr);
}

}  // namespace tlpk
