// tlpk_ipm.cpp -- C ABI of the device-resident interior-point vectors (include/tlpk.h, section
// "Device-resident HSD iterate").
//
// With the drop-in KKT interface (tlpk_update / tlpk_solve) every solve moves 16 (m + n) bytes over
// PCIe and the host builds every right-hand side.  Here the iterate, the residuals, the right-hand
// sides and the search directions of Tulip's homogeneous self-dual loop live in HBM; a call runs one
// routine of /root/reference/src/IPM/HSD/{HSD.jl, step.jl} on the device and returns the handful of
// scalars the host logic needs (norms, dot products, step lengths).  The host keeps tau, kappa, the
// regularisation scalars and the control flow (tulip.jl_amd/hsd_device.py mirrors HSD.jl:203-350).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "tlpk_handle.hpp"
#include "tlpk_ipm.hpp"

struct IpmState {
    IpmVecs v{};
    IpmDir D[2]{};                  // D[cur] = accepted direction, D[1 - cur] = candidate of the centrality corrector
    int cur = 0;
    double *partials[2] = {nullptr, nullptr};     // per-block partial results of two concurrent reductions
    double *d_out = nullptr;        // finalised scalars on the device
    double *h_out = nullptr;        // ... and in pinned host memory
    int pre_blocks = 0;
    // multi-device parent only: the linking rows (user indices, ascending), b on them, staging for the shards' partial rp
    std::vector<i64> link_rows; std::vector<double> b_link; double *h_link = nullptr; i64 link_lo = 0, link_hi = 0;
};

// ---- one device or several ---------------------------------------------------------------------------------------------------
// On a tlpk_create_multi handle (block-angular LP; K1, or K2 -- then the variable nodes of the replicated root front count as the
// lead shard's columns) every shard holds the SUB-LP of its diagonal blocks in vectors of the job's
// full length: its own columns with their costs and bounds (the other columns are empty: cost 0, no bounds, no entries of A), its
// block rows of b, and the linking rows -- b on the lead shard only, A restricted to the shard's columns.  Every kernel of
// ipm_kernels.hip then computes, unchanged, the shard's share of each sum / maximum / minimum (an empty column or row contributes
// the neutral element), the host combines the shards' scalars in shard order, and three things cross the shards:
//   * the KKT solves: split-phase, EVERY shard adds its xi_p on the linking rows (its partial residual; b only on the lead), the
//     library's reduction of the root right-hand side completes the rows; solutions stay shard-resident (own columns / block
//     rows, linking rows replicated -- y, dy on them are updated identically everywhere);
//   * the factorisation: the usual reduction of the root panel;
//   * |rp|inf and |A x|inf on the linking rows (tlpk_ipm_residuals): the shards' partial rp of those rows are summed on the host.
// A single-device handle is the same code with one shard.
namespace {

struct Shards { tlpk_handle *c[MAX_DEVICES]; int n = 0; bool multi = false; };

int ipm_shards(tlpk_handle *h, Shards &sh, bool need_loaded = true) {
    if (!h) return TLPK_BADARG;
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (h->opt.nranks > 1) { h->last_error = "the device-resident IPM vectors need a single-rank or a tlpk_create_multi handle (a sharded handle's reductions belong to its caller)"; return TLPK_BADARG; }
    if (!h->sub.empty()) {
        sh.multi = true; sh.n = (int)h->sub.size();
        for (int r = 0; r < sh.n; ++r) sh.c[r] = h->sub[(size_t)r];
    } else { sh.multi = false; sh.n = 1; sh.c[0] = h; }
    if (need_loaded && !h->ipm) { h->last_error = "tlpk_ipm_load has not been called"; return TLPK_BADARG; }
    return TLPK_OK;
}
int fail_from(tlpk_handle *h, tlpk_handle *c, int rc) { if (c != h) h->last_error = c->last_error; return rc; }

// after every shard has finalised `count` scalars at d_out: copy them to the host, wait, combine in shard order
// (slots [0, nsum) sums, then nmax maxima, then nmin minima) into res
int gather(tlpk_handle *h, Shards &sh, int count, int nsum, int nmax, double *res) {
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipMemcpyAsync(s.h_out, s.d_out, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream));
    }
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r];
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipStreamSynchronize(c->stream));
        HIPCHK(h, hipGetLastError());
    }
    for (int k = 0; k < count; ++k) {
        double v = sh.c[0]->ipm->h_out[k];
        for (int r = 1; r < sh.n; ++r) {
            const double w = sh.c[r]->ipm->h_out[k];
            v = (k < nsum) ? v + w : (k < nsum + nmax ? std::fmax(v, w) : std::fmin(v, w));
        }
        res[k] = v;
    }
    return TLPK_OK;
}
// slots laid out in two groups of IPM_SLOTS (two concurrent reductions): group g has its own (nsum, nmax)
int gather2(tlpk_handle *h, Shards &sh, int nsum0, int nmax0, int cnt0, int nsum1, int nmax1, int cnt1, double *res) {
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipMemcpyAsync(s.h_out, s.d_out, (size_t)(IPM_SLOTS + cnt1) * 8, hipMemcpyDeviceToHost, c->stream));
    }
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r];
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipStreamSynchronize(c->stream));
        HIPCHK(h, hipGetLastError());
    }
    for (int g = 0; g < 2; ++g) {
        const int off = g * IPM_SLOTS, cnt = g ? cnt1 : cnt0, ns = g ? nsum1 : nsum0, nm = g ? nmax1 : nmax0;
        for (int k = 0; k < cnt; ++k) {
            double v = sh.c[0]->ipm->h_out[off + k];
            for (int r = 1; r < sh.n; ++r) {
                const double w = sh.c[r]->ipm->h_out[off + k];
                v = (k < ns) ? v + w : (k < ns + nm ? std::fmax(v, w) : std::fmin(v, w));
            }
            res[off + k] = v;
        }
    }
    return TLPK_OK;
}
int sync_all(tlpk_handle *h, Shards &sh) {
    for (int r = 0; r < sh.n; ++r) { const int rc = tlpk_sync(sh.c[r]); if (rc != TLPK_OK) return fail_from(h, sh.c[r], rc); }
    return TLPK_OK;
}
// KKT.update! with the theta / regularisation vectors every shard has just written
int update_all(tlpk_handle *h, Shards &sh) {
    if (!sh.multi) return tlpk_update_device(h, h->d_theta, h->d_regP, h->d_regD);
    return multi_update_resident(h);
}
// KKT.solve! per shard: pick(state) -> {dx, dy, xi_p, xi_d}
template <class F>
int solve_all(tlpk_handle *h, Shards &sh, F &&pick) {
    double *dx[MAX_DEVICES], *dy[MAX_DEVICES]; const double *xp[MAX_DEVICES], *xd[MAX_DEVICES];
    for (int r = 0; r < sh.n; ++r) pick(*sh.c[r]->ipm, dx[r], dy[r], xp[r], xd[r]);
    if (!sh.multi) return tlpk_solve_device(h, dx[0], dy[0], xp[0], xd[0]);
    return multi_solve_resident(h, dx, dy, xp, xd);
}

}  // namespace

void ipm_free(tlpk_handle *h) {
    if (!h || !h->ipm) return;
    if (h->ipm->h_out) hipHostFree(h->ipm->h_out);
    if (h->ipm->h_link) hipHostFree(h->ipm->h_link);
    delete h->ipm;                  // device vectors are in h->allocs
    h->ipm = nullptr;
}

extern "C" {

// mask = nullptr: the whole LP on handle c (a single-device handle).  Otherwise c is a shard of a multi-device handle and takes the
// sub-LP of the columns / rows it owns (see the comment at the top).
static int ipm_load_impl(tlpk_handle *h, const double *b, const double *c, const double *l, const double *u, bool shard);

int tlpk_ipm_load(tlpk_handle *h, const double *b, const double *c, const double *l, const double *u) {
    // a failed load leaves no half-initialised state behind: later tlpk_ipm_* calls report "not loaded", a retry is possible
    Shards sh;
    if (int rc = ipm_shards(h, sh, false)) return rc;
    if (!b || !c || !l || !u) return TLPK_BADARG;
    if (h->ipm) { h->last_error = "tlpk_ipm_load called twice on one handle"; return TLPK_BADARG; }
    int rc = TLPK_OK;
    if (!sh.multi) rc = ipm_load_impl(h, b, c, l, u, false);
    else {
        for (int r = 0; r < sh.n && rc == TLPK_OK; ++r) {
            rc = ipm_load_impl(sh.c[r], b, c, l, u, true);
            if (rc != TLPK_OK) h->last_error = sh.c[r]->last_error;
        }
        if (rc == TLPK_OK) {
            IpmState *sp = new (std::nothrow) IpmState();           // the parent's state: what the host needs for the linking rows
            if (!sp) rc = TLPK_OOM;
            else {
                h->ipm = sp;
                const Symbolic &S0 = sh.c[0]->S;
                const bool k2 = S0.system == 1;
                const i64 mm = k2 ? S0.k2_m : S0.m, off = k2 ? S0.k2_n : 0;
                for (i64 i = 0; i < mm; ++i) if (S0.row_local[(size_t)(off + i)] == 2) { sp->link_rows.push_back(i); sp->b_link.push_back(b[i]); }
                if (!sp->link_rows.empty()) {
                    sp->link_lo = sp->link_rows.front(); sp->link_hi = sp->link_rows.back() + 1;
                    if (hipHostMalloc((void **)&sp->h_link, (size_t)(sp->link_hi - sp->link_lo) * (size_t)sh.n * 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); rc = TLPK_OOM; }
                }
            }
        }
    }
    if (rc != TLPK_OK) { ipm_free(h); if (sh.multi) for (int r = 0; r < sh.n; ++r) ipm_free(sh.c[r]); }
    return rc;
}

static int ipm_load_impl(tlpk_handle *h, const double *b, const double *c, const double *l, const double *u, bool shard) {
    HIPCHK(h, hipSetDevice(h->device));
    const bool k2 = h->S.system == 1;
    const i64 m = k2 ? h->S.k2_m : h->S.m, n = k2 ? h->S.k2_n : h->S.n;
    IpmState *sp = new (std::nothrow) IpmState();
    if (!sp) return TLPK_OOM;
    h->ipm = sp;
    IpmState &s = *sp;
    IpmVecs &v = s.v;
    v.m = m; v.n = n; v.row_skip = nullptr;
    int rc;
    const Symbolic &S = h->S;
    // ownership of columns (variables) and rows (constraints) on a shard: K1 -- the column / row maps of the analyse phase; K2 -- the
    // node map (variable nodes 0 .. n-1, constraint nodes n .. n+m-1): a node of the replicated root front belongs to the lead shard
    auto own_col = [&](i64 j) { if (!shard) return true; if (!k2) return S.col_local[(size_t)j] != 0; const char nl = S.row_local[(size_t)j]; return nl == 1 || (nl == 2 && h->opt.rank == 0); };
    auto row_kind = [&](i64 i) -> char { return k2 ? S.row_local[(size_t)(n + i)] : S.row_local[(size_t)i]; };      // 0 other shard, 1 own, 2 linking (replicated)
    if (!shard && !k2) { v.Ap = h->d.Ap; v.Ai = h->d.Ai; v.Ax = h->d.Ax; v.Tp = h->d.Tp; v.Tj = h->d.Tj; v.Tx = h->d.Tx; }
    else {
        // host copy of A, column-major: K1 -- the analyse phase's copy; K2 -- rebuilt from the incidence matrix of the augmented system
        // it holds (column p = entry p of A: 1 on variable node j, A[i,j] on constraint node n + i, in A's column-major entry order).
        // A shard keeps the columns it owns (the others are empty).  Then the row-major copy.
        std::vector<i64> ap((size_t)n + 1, 0), tp((size_t)m + 1, 0);
        std::vector<i32> ai, tj; std::vector<double> ax, tx;
        if (!k2) {
            for (i64 j = 0; j < n; ++j) {
                if (own_col(j)) for (i64 q = S.Ap[(size_t)j]; q < S.Ap[(size_t)j + 1]; ++q) { ai.push_back(S.Ai[(size_t)q]); ax.push_back(S.Ax[(size_t)q]); }
                ap[(size_t)j + 1] = (i64)ai.size();
            }
        } else {
            const i64 nnz = S.n;
            i64 jprev = 0;
            for (i64 p = 0; p < nnz; ++p) {
                if (S.Ap[(size_t)p + 1] - S.Ap[(size_t)p] != 2) { h->last_error = "K2 incidence matrix: unexpected column"; return TLPK_INTERNAL; }
                const i64 q = S.Ap[(size_t)p];
                const i32 a = S.Ai[(size_t)q], b2 = S.Ai[(size_t)q + 1];
                const bool afirst = a < (i32)n;                   // the variable node is the smaller index
                const i64 j = afirst ? a : b2; const i32 i = (afirst ? b2 : a) - (i32)n;
                if (j < jprev) { h->last_error = "K2 incidence matrix: columns out of order"; return TLPK_INTERNAL; }
                for (; jprev < j; ++jprev) ap[(size_t)jprev + 1] = (i64)ai.size();
                if (own_col(j)) { ai.push_back(i); ax.push_back(S.Ax[(size_t)q + (afirst ? 1 : 0)]); }
            }
            for (; jprev < n; ++jprev) ap[(size_t)jprev + 1] = (i64)ai.size();
        }
        for (size_t q = 0; q < ai.size(); ++q) ++tp[(size_t)ai[q] + 1];
        for (i64 i = 0; i < m; ++i) tp[(size_t)i + 1] += tp[(size_t)i];
        tj.resize(ai.size()); tx.resize(ai.size());
        { std::vector<i64> cur(tp.begin(), tp.end() - 1);
          for (i64 j = 0; j < n; ++j) for (i64 q = ap[(size_t)j]; q < ap[(size_t)j + 1]; ++q) { const i64 c2 = cur[(size_t)ai[(size_t)q]]++; tj[(size_t)c2] = (i32)j; tx[(size_t)c2] = ax[(size_t)q]; } }
        i64 *dp; i32 *di; double *dxv;
        if ((rc = dev_upload(h, &dp, ap)) != TLPK_OK) return rc; v.Ap = dp;
        if ((rc = dev_upload(h, &di, ai)) != TLPK_OK) return rc; v.Ai = di;
        if ((rc = dev_upload(h, &dxv, ax)) != TLPK_OK) return rc; v.Ax = dxv;
        if ((rc = dev_upload(h, &dp, tp)) != TLPK_OK) return rc; v.Tp = dp;
        if ((rc = dev_upload(h, &di, tj)) != TLPK_OK) return rc; v.Tj = di;
        if ((rc = dev_upload(h, &dxv, tx)) != TLPK_OK) return rc; v.Tx = dxv;
        if (shard) {
            char *dc;
            std::vector<char> skip((size_t)m);
            for (i64 i = 0; i < m; ++i) skip[(size_t)i] = row_kind(i) == 2;      // linking rows: partial sums, no part in the maxima
            if ((rc = dev_upload(h, &dc, skip)) != TLPK_OK) return rc; v.row_skip = dc;
        }
    }
    // problem data: b, c, l .* lflag, u .* uflag, flags (ipmdata.jl:46-47); a shard: its sub-LP
    std::vector<double> lz((size_t)n), uz((size_t)n), lf((size_t)n), uf((size_t)n), bb, cc;
    for (i64 j = 0; j < n; ++j) {
        const bool own = own_col(j);
        const bool fl = own && std::isfinite(l[j]), fu = own && std::isfinite(u[j]);
        lf[(size_t)j] = fl ? 1.0 : 0.0; uf[(size_t)j] = fu ? 1.0 : 0.0;
        lz[(size_t)j] = fl ? l[j] : 0.0; uz[(size_t)j] = fu ? u[j] : 0.0;
    }
    if (shard) {
        bb.assign((size_t)m, 0.0); cc.assign((size_t)n, 0.0);
        for (i64 i = 0; i < m; ++i) { const char rl = row_kind(i); if (rl == 1 || (rl == 2 && h->opt.rank == 0)) bb[(size_t)i] = b[i]; }
        for (i64 j = 0; j < n; ++j) if (own_col(j)) cc[(size_t)j] = c[j];
        b = bb.data(); c = cc.data();
    }
    double *p;
#define UPV(dst, ptr, len) do { if ((rc = dev_alloc(h, &p, (len))) != TLPK_OK) return rc; if ((len) > 0) HIPCHK(h, hipMemcpy(p, (ptr), (size_t)(len) * 8, hipMemcpyHostToDevice)); dst = p; } while (0)
    UPV(v.b, b, m); UPV(v.c, c, n); UPV(v.lz, lz.data(), n); UPV(v.uz, uz.data(), n); UPV(v.lflag, lf.data(), n); UPV(v.uflag, uf.data(), n);
#undef UPV
#define ALV(dst, len) do { if ((rc = dev_alloc(h, &p, (len))) != TLPK_OK) return rc; HIPCHK(h, hipMemset(p, 0, (size_t)std::max<i64>((len), 1) * 8)); dst = p; } while (0)
    ALV(v.x, n); ALV(v.xl, n); ALV(v.xu, n); ALV(v.zl, n); ALV(v.zu, n); ALV(v.y, m);
    ALV(v.rp, m); ALV(v.rl, n); ALV(v.ru, n); ALV(v.rd, n); ALV(v.thl, n); ALV(v.thu, n); ALV(v.hx, n); ALV(v.hy, m); ALV(v.hxid, n);
    ALV(v.xil, n); ALV(v.xiu, n); ALV(v.xzl, n); ALV(v.xzu, n); ALV(v.xid, n); ALV(v.xip, m);
    for (int k = 0; k < 2; ++k) { ALV(s.D[k].x, n); ALV(s.D[k].xl, n); ALV(s.D[k].xu, n); ALV(s.D[k].zl, n); ALV(s.D[k].zu, n); ALV(s.D[k].y, m); }
    ALV(s.partials[0], (i64)IPM_BLOCKS * IPM_SLOTS); ALV(s.partials[1], (i64)IPM_BLOCKS * IPM_SLOTS);
    ALV(s.d_out, 2 * IPM_SLOTS);
#undef ALV
    HIPCHK(h, hipHostMalloc((void **)&s.h_out, 2 * IPM_SLOTS * sizeof(double), hipHostMallocDefault));
    ipm_launch_init(h->stream, v);                                       // HSD.jl:238-247
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return TLPK_OK;
}

int tlpk_ipm_reset(tlpk_handle *h) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r];
        HIPCHK(h, hipSetDevice(c->device));
        ipm_launch_init(c->stream, c->ipm->v);
    }
    return TLPK_OK;
}

/* HSD.jl:77-128 + the quantities of HSD.jl:136-196.  out[13]:
 *  0 |rp|inf  1 |rl|inf  2 |ru|inf  3 |rd|inf  4 c'x  5 b'y  6 lz'zl  7 uz'zu  8 xl'zl + xu'zu
 *  9 |A x|inf  10 |(x - xl) lflag|inf  11 |(x + xu) uflag|inf  12 |A'y + zl lflag - zu uflag|inf */
int tlpk_ipm_residuals(tlpk_handle *h, double tau, double *out) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!out) return TLPK_BADARG;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        const int nbc = ipm_launch_res_cols(c->stream, s.v, tau, s.partials[0]);
        const int nbr = ipm_launch_res_rows(c->stream, s.v, tau, s.partials[1]);
        ipm_launch_finalize(c->stream, nbc, 4, 6, 0, s.partials[0], s.d_out);
        ipm_launch_finalize(c->stream, nbr, 1, 2, 0, s.partials[1], s.d_out + IPM_SLOTS);
        if (sh.multi && h->ipm->h_link) {       // the shard's partial rp on the linking rows
            IpmState &ps = *h->ipm; const i64 len = ps.link_hi - ps.link_lo;
            HIPCHK(h, hipMemcpyAsync(ps.h_link + (size_t)r * (size_t)len, s.v.rp + ps.link_lo, (size_t)len * 8, hipMemcpyDeviceToHost, c->stream));
        }
    }
    double q[2 * IPM_SLOTS];
    if (int rc = gather2(h, sh, 4, 6, 10, 1, 2, 3, q)) return rc;
    const double *a = q, *r = q + IPM_SLOTS;
    double rp_inf = r[1], ax_inf = r[2];
    if (sh.multi && h->ipm->h_link) {
        // linking rows: rp = sum of the shards' partial rows (shard order), A x = tau b - rp
        IpmState &ps = *h->ipm; const i64 len = ps.link_hi - ps.link_lo;
        for (size_t k = 0; k < ps.link_rows.size(); ++k) {
            const i64 off = ps.link_rows[k] - ps.link_lo;
            double v = ps.h_link[off];
            for (int s2 = 1; s2 < sh.n; ++s2) v += ps.h_link[(size_t)s2 * (size_t)len + (size_t)off];
            rp_inf = std::fmax(rp_inf, std::fabs(v));
            ax_inf = std::fmax(ax_inf, std::fabs(tau * ps.b_link[k] - v));
        }
    }
    out[0] = rp_inf; out[1] = a[4]; out[2] = a[5]; out[3] = a[6]; out[4] = a[0]; out[5] = r[0]; out[6] = a[1]; out[7] = a[2];
    out[8] = a[3]; out[9] = ax_inf; out[10] = a[7]; out[11] = a[8]; out[12] = a[9];
    return TLPK_OK;
}

/* step.jl:24-51: theta_inv from the iterate, uniform regularisations, KKT.update!.  TLPK_NOT_POSDEF is
 * the PosDefException of the retry loop: call again with larger regularisations. */
int tlpk_ipm_factor(tlpk_handle *h, double regP, double regD) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r];
        HIPCHK(h, hipSetDevice(c->device));
        ipm_launch_theta(c->stream, c->ipm->v, c->d_theta, c->d_regP, c->d_regD, regP, regD);
    }
    return update_all(h, sh);
}

/* step.jl:56-76: solve the h-system (xi_p = b, xi_d = c - th_l lz - th_u uz); hx, hy stay on the device.
 * out[0] = lz'(lz th_l) + uz'(uz th_u) - (c + th_l lz + th_u uz)'hx + b'hy   (the host adds kappa/tau + regG) */
int tlpk_ipm_hsolve(tlpk_handle *h, double *out) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!out) return TLPK_BADARG;
    for (int r = 0; r < sh.n; ++r) { HIPCHK(h, hipSetDevice(sh.c[r]->device)); ipm_launch_hrhs(sh.c[r]->stream, sh.c[r]->ipm->v); }
    int rc = solve_all(h, sh, [](IpmState &s, double *&dx, double *&dy, const double *&xp, const double *&xd) { dx = s.v.hx; dy = s.v.hy; xp = s.v.b; xd = s.v.hxid; });
    if (rc != TLPK_OK) return rc;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        const int nb = ipm_launch_hdots(c->stream, s.v, s.partials[0]);
        ipm_launch_finalize(c->stream, nb, 2, 0, 0, s.partials[0], s.d_out);
    }
    double q[IPM_SLOTS];
    if ((rc = gather(h, sh, 2, 2, 0, q)) != TLPK_OK) return rc;
    if ((rc = sync_all(h, sh)) != TLPK_OK) return rc;
    out[0] = q[0] + q[1];
    return TLPK_OK;
}

/* step.jl:325-364 (first half of compute_higher_corrector): targets from the accepted direction;
 * out[0] = sum(vl), out[1] = sum(vu); the host adds the tau-kappa term and forms delta. */
int tlpk_ipm_targets(tlpk_handle *h, double a_, double mu_l, double mu_u, double *out) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!out) return TLPK_BADARG;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        const int nb = ipm_launch_targets(c->stream, s.v, s.D[s.cur], a_, a_, mu_l, mu_u, s.partials[0]);
        ipm_launch_finalize(c->stream, nb, 2, 0, 0, s.partials[0], s.d_out);
    }
    double q[IPM_SLOTS];
    if (int rc = gather(h, sh, 2, 2, 0, q)) return rc;
    out[0] = q[0]; out[1] = q[1];
    return TLPK_OK;
}

/* step.jl:198-266 (solve_newton_system) + step.jl:294-306 (max step) for one right-hand side:
 *   mode 0 predictor, mode 1 corrector, mode 2 centrality corrector (after tlpk_ipm_targets).
 * sc[8] = { tau, kappa, h0, xi_g, xi_tk, eta, gamma*mu, delta }.
 * Modes 0 / 1 write the accepted direction, mode 2 writes the candidate (Dc = solution + accepted direction).
 * out[3] = { dtau, dkappa, largest step to the boundary over xl, xu, zl, zu (inf if none) } of the written
 * direction; for mode 2 dtau / dkappa are those of the solution alone (the host adds the accepted ones). */
int tlpk_ipm_newton(tlpk_handle *h, int mode, const double *sc, double *out) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!sc || !out || mode < 0 || mode > 2) return TLPK_BADARG;
    const double tau = sc[0], kappa = sc[1], h0 = sc[2], xi_g = sc[3], xi_tk = sc[4], eta = sc[5], gmu = sc[6], delta = sc[7];
    int nb[MAX_DEVICES];
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        nb[r] = ipm_launch_newton_pre(c->stream, s.v, s.D[s.cur], mode, eta, gmu, delta, s.partials[0]);
    }
    int rc = solve_all(h, sh, [mode](IpmState &s, double *&dx, double *&dy, const double *&xp, const double *&xd) {
        const IpmDir &dst = (mode == 2) ? s.D[1 - s.cur] : s.D[s.cur]; dx = dst.x; dy = dst.y; xp = s.v.xip; xd = s.v.xid; });
    if (rc != TLPK_OK) return rc;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        const IpmDir &dst = (mode == 2) ? s.D[1 - s.cur] : s.D[s.cur];
        HIPCHK(h, hipSetDevice(c->device));
        ipm_launch_newton_dots(c->stream, s.v, dst, nb[r], s.partials[0]);
        ipm_launch_finalize(c->stream, nb[r], 6, 0, 0, s.partials[0], s.d_out);
    }
    double q[IPM_SLOTS];
    if ((rc = gather(h, sh, 6, 6, 0, q)) != TLPK_OK) return rc;
    if ((rc = sync_all(h, sh)) != TLPK_OK) return rc;
    // step.jl:232-246
    const double xi_g_ = xi_g + xi_tk / tau - q[0] + q[1] - q[2] - q[3];
    const double dtau = (xi_g_ + q[4] - q[5]) / h0;
    const double dkappa = (xi_tk - kappa * dtau) / tau;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        const IpmDir &acc = s.D[s.cur];
        const IpmDir &dst = (mode == 2) ? s.D[1 - s.cur] : s.D[s.cur];
        HIPCHK(h, hipSetDevice(c->device));
        const int nb2 = ipm_launch_newton_post(c->stream, s.v, dst, acc, mode == 2 ? 1 : 0, dtau, s.partials[1]);
        ipm_launch_finalize(c->stream, nb2, 0, 0, 2, s.partials[1], s.d_out);
    }
    if ((rc = gather(h, sh, 2, 0, 0, q)) != TLPK_OK) return rc;
    out[0] = dtau; out[1] = dkappa; out[2] = std::fmin(q[0], q[1]);    // one step length for both sides
    return TLPK_OK;
}

/* step.jl:24-94 in ONE call: tlpk_ipm_factor (theta_inv from the iterate, uniform regularisations, KKT.update!) WITHOUT the wait for its
 * status, followed by tlpk_ipm_hsolve_newton: the block-level forward sweeps of the paired solve overlap the factorisation of the root
 * (linking) front (tlpk_update_device_async).  TLPK_NOT_POSDEF is returned where tlpk_ipm_factor would have returned it -- the
 * speculative solve has only written direction / h-system buffers -- and the caller retries with larger regularisations (step.jl:35-51).
 * Multi-device handles: the blocking factorisation, then tlpk_ipm_hsolve_newton. */
int tlpk_ipm_factor_hsolve_newton(tlpk_handle *h, double regP, double regD, const double *sc, double *out) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!sc || !out) return TLPK_BADARG;
    if (sh.multi) { const int rc = tlpk_ipm_factor(h, regP, regD); return rc != TLPK_OK ? rc : tlpk_ipm_hsolve_newton(h, sc, out); }
    HIPCHK(h, hipSetDevice(h->device));
    ipm_launch_theta(h->stream, h->ipm->v, h->d_theta, h->d_regP, h->d_regD, regP, regD);
    const int rc = tlpk_update_device_async(h, h->d_theta, h->d_regP, h->d_regD);
    if (rc != TLPK_OK) return rc;
    return tlpk_ipm_hsolve_newton(h, sc, out);
}

/* step.jl:56-94 in ONE call: the h-system (step.jl:56-76) and the predictor's Newton system (mode 0 of tlpk_ipm_newton) are
 * independent right-hand sides against the same factor -- they share one pass over L (tlpk_solve2_device: the sweeps are bound by
 * the bytes of L; multi-device handles: the split-phase pair, two reductions of root right-hand sides).  sc[8] as for tlpk_ipm_newton, except sc[2] = regG (h0 = dot products +
 * kappa / tau + regG is formed here).  out[4] = { dtau, dkappa, largest step to the boundary, h0 }.  Same arithmetic as
 * tlpk_ipm_hsolve + tlpk_ipm_newton(mode 0): bit-identical vectors and scalars. */
int tlpk_ipm_hsolve_newton(tlpk_handle *h, const double *sc, double *out) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!sc || !out) return TLPK_BADARG;
    const double tau = sc[0], kappa = sc[1], regG = sc[2], xi_g = sc[3], xi_tk = sc[4], eta = sc[5], gmu = sc[6], delta = sc[7];
    int nb[MAX_DEVICES];
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        ipm_launch_hrhs(c->stream, s.v);
        nb[r] = ipm_launch_newton_pre(c->stream, s.v, s.D[s.cur], 0, eta, gmu, delta, s.partials[0]);
    }
    int rc;
    if (!sh.multi) {
        IpmState &s = *h->ipm; const IpmDir &dst = s.D[s.cur];
        rc = tlpk_solve2_device(h, s.v.hx, s.v.hy, s.v.b, s.v.hxid, dst.x, dst.y, s.v.xip, s.v.xid);
    } else {
        double *dx0[MAX_DEVICES], *dy0[MAX_DEVICES], *dx1[MAX_DEVICES], *dy1[MAX_DEVICES];
        const double *xp0[MAX_DEVICES], *xd0[MAX_DEVICES], *xp1[MAX_DEVICES], *xd1[MAX_DEVICES];
        for (int r = 0; r < sh.n; ++r) {
            IpmState &s = *sh.c[r]->ipm;
            dx0[r] = s.v.hx; dy0[r] = s.v.hy; xp0[r] = s.v.b; xd0[r] = s.v.hxid;
            dx1[r] = s.D[s.cur].x; dy1[r] = s.D[s.cur].y; xp1[r] = s.v.xip; xd1[r] = s.v.xid;
        }
        rc = multi_solve2_resident(h, dx0, dy0, xp0, xd0, dx1, dy1, xp1, xd1);
    }
    if (rc != TLPK_OK) return rc;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm; const IpmDir &dst = s.D[s.cur];
        HIPCHK(h, hipSetDevice(c->device));
        const int nbh = ipm_launch_hdots(c->stream, s.v, s.partials[1]);
        ipm_launch_finalize(c->stream, nbh, 2, 0, 0, s.partials[1], s.d_out + IPM_SLOTS);
        ipm_launch_newton_dots(c->stream, s.v, dst, nb[r], s.partials[0]);
        ipm_launch_finalize(c->stream, nb[r], 6, 0, 0, s.partials[0], s.d_out);
    }
    double q[2 * IPM_SLOTS];
    if ((rc = gather2(h, sh, 6, 0, 6, 2, 0, 2, q)) != TLPK_OK) return rc;
    if ((rc = sync_all(h, sh)) != TLPK_OK) return rc;
    const double h0 = ((q[IPM_SLOTS] + q[IPM_SLOTS + 1]) + kappa / tau) + regG;      // the association of hsd_device.py / HSD/step.jl:69-76
    const double xi_g_ = xi_g + xi_tk / tau - q[0] + q[1] - q[2] - q[3];
    const double dtau = (xi_g_ + q[4] - q[5]) / h0;
    const double dkappa = (xi_tk - kappa * dtau) / tau;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm; const IpmDir &dst = s.D[s.cur];
        HIPCHK(h, hipSetDevice(c->device));
        const int nb2 = ipm_launch_newton_post(c->stream, s.v, dst, dst, 0, dtau, s.partials[1]);
        ipm_launch_finalize(c->stream, nb2, 0, 0, 2, s.partials[1], s.d_out);
    }
    if ((rc = gather(h, sh, 2, 0, 0, q)) != TLPK_OK) return rc;
    out[0] = dtau; out[1] = dkappa; out[2] = std::fmin(q[0], q[1]); out[3] = h0;
    return TLPK_OK;
}

/* step.jl:112-118: the candidate of the last mode-2 call becomes the accepted direction */
int tlpk_ipm_accept(tlpk_handle *h) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    for (int r = 0; r < sh.n; ++r) sh.c[r]->ipm->cur = 1 - sh.c[r]->ipm->cur;
    return TLPK_OK;
}

/* step.jl:139-148: pt += alpha * D; out[0] = xl'zl + xu'zu of the new point (mu numerator, point.jl:45-48) */
static int advance_all(tlpk_handle *h, double ap, double ad, double *out) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!out) return TLPK_BADARG;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        const int nb = ipm_launch_advance(c->stream, s.v, s.D[s.cur], ap, ad, s.partials[0]);
        ipm_launch_finalize(c->stream, nb, 1, 0, 0, s.partials[0], s.d_out);
    }
    double q[IPM_SLOTS];
    if (int rc = gather(h, sh, 1, 1, 0, q)) return rc;
    out[0] = q[0];
    return TLPK_OK;
}
int tlpk_ipm_advance(tlpk_handle *h, double alpha, double *out) { return advance_all(h, alpha, alpha, out); }

/* download one vector of the iterate: what = 0 x, 1 xl, 2 xu, 3 zl, 4 zu (length n), 5 y (length m) */
int tlpk_ipm_get(tlpk_handle *h, int what, double *host, int64_t len) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!host || what < 0 || what > 5) return TLPK_BADARG;
    const IpmVecs &v0 = sh.c[0]->ipm->v;
    const int64_t need = (what == 5) ? v0.m : v0.n;
    if (len != need) { h->last_error = "tlpk_ipm_get: wrong length"; return TLPK_BADARG; }
    std::vector<double> tmp;
    if (sh.multi) tmp.resize((size_t)std::max<int64_t>(need, 1));
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; const IpmVecs &v = c->ipm->v;
        const double *src[6] = {v.x, v.xl, v.xu, v.zl, v.zu, v.y};
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipStreamSynchronize(c->stream));
        if (need <= 0) continue;
        if (!sh.multi) { HIPCHK(h, hipMemcpy(host, src[what], (size_t)need * 8, hipMemcpyDeviceToHost)); continue; }
        // every entry from the shard that owns it (linking rows: the lead)
        HIPCHK(h, hipMemcpy(tmp.data(), src[what], (size_t)need * 8, hipMemcpyDeviceToHost));
        const bool k2 = c->S.system == 1;
        const std::vector<char> &rl = c->S.row_local;
        if (what == 5) { const int64_t off = k2 ? c->S.k2_n : 0; for (int64_t i = 0; i < need; ++i) { const char q = rl[(size_t)(off + i)]; if (q == 1 || (q == 2 && r == 0)) host[i] = tmp[(size_t)i]; } }
        else if (k2) { for (int64_t j = 0; j < need; ++j) { const char q = rl[(size_t)j]; if (q == 1 || (q == 2 && r == 0)) host[j] = tmp[(size_t)j]; } }
        else { const std::vector<char> &cl = c->S.col_local; for (int64_t j = 0; j < need; ++j) if (cl[(size_t)j]) host[j] = tmp[(size_t)j]; }
    }
    return TLPK_OK;
}

/* ---- Mehrotra predictor-corrector with the iterate in HBM (MPC/MPC.jl, MPC/step.jl) ------------------------
 * Shares the vectors, tlpk_ipm_load / residuals (tau = 1) / factor / accept / get with the HSD entry points. */

/* MPC.jl:353-410: starting point.  One factorisation of A A' + 1e-6 I, two solves with a zero half of the
 * right-hand side, shifts to positive coordinates, balanced products.  out[0] = xl'zl + xu'zu. */
int tlpk_mpc_start(tlpk_handle *h, double *out) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!out) return TLPK_BADARG;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r];
        HIPCHK(h, hipSetDevice(c->device));
        mpc_launch_fill(c->stream, c->ipm->v, c->d_theta, c->d_regP, c->d_regD);
    }
    int rc = update_all(h, sh);
    if (rc != TLPK_OK) return rc;
    rc = solve_all(h, sh, [](IpmState &s, double *&dx, double *&dy, const double *&xp, const double *&xd) { dx = s.D[0].x; dy = s.v.y; xp = s.v.xip; xd = s.v.c; });      // y  (xip == 0 here)
    if (rc != TLPK_OK) return rc;
    rc = solve_all(h, sh, [](IpmState &s, double *&dx, double *&dy, const double *&xp, const double *&xd) { dx = s.v.x; dy = s.D[0].y; xp = s.v.b; xd = s.v.xid; });      // x  (xid == 0 here)
    if (rc != TLPK_OK) return rc;
    double q[IPM_SLOTS];
    auto stage = [&](int st, double a, double b, int nsum, int nmin) -> int {
        for (int r = 0; r < sh.n; ++r) {
            tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
            HIPCHK(h, hipSetDevice(c->device));
            const int nb = mpc_launch_start(c->stream, s.v, st, a, b, s.partials[0]);
            ipm_launch_finalize(c->stream, nb, nsum, 0, nmin, s.partials[0], s.d_out);
        }
        return gather(h, sh, nsum + nmin, nsum, 0, q);
    };
    if ((rc = stage(1, 0.0, 0.0, 0, 2)) != TLPK_OK) return rc;
    if ((rc = sync_all(h, sh)) != TLPK_OK) return rc;
    const double dxs = 1.0 + std::fmax(0.0, std::fmax(-1.5 * q[0], -1.5 * q[1]));
    if ((rc = stage(2, dxs, 0.0, 0, 2)) != TLPK_OK) return rc;
    const double dzs = 1.0 + std::fmax(0.0, std::fmax(-1.5 * q[0], -1.5 * q[1]));
    if ((rc = stage(3, dzs, 0.0, 3, 0)) != TLPK_OK) return rc;
    const double mu = q[0], ddx = mu / (2.0 * q[1]), ddz = mu / (2.0 * q[2]);
    if ((rc = stage(4, ddx, ddz, 1, 0)) != TLPK_OK) return rc;
    out[0] = q[0];
    for (int r = 0; r < sh.n; ++r) sh.c[r]->ipm->cur = 0;
    return TLPK_OK;
}

/* MPC/step.jl:164-217 (solve_newton_system + max_step_length_pd) for one right-hand side:
 *   mode 0 predictor (xi = residuals, complementarity -x z), mode 1 corrector (same residuals, sigma mu - x z - dx dz
 *   with the predictor direction, which it overwrites; gmu = sigma mu), mode 2 centrality corrector (zero residuals,
 *   the targets of tlpk_mpc_targets; writes the candidate = solution + accepted direction).
 * out[2] = largest primal step (over xl, xu) and largest dual step (over zl, zu) to the boundary, inf if none. */
int tlpk_mpc_newton(tlpk_handle *h, int mode, double gmu, double *out) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!out || mode < 0 || mode > 2) return TLPK_BADARG;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        ipm_launch_newton_pre(c->stream, s.v, s.D[s.cur], mode, 1.0, gmu, 0.0, s.partials[0]);
    }
    int rc = solve_all(h, sh, [mode](IpmState &s, double *&dx, double *&dy, const double *&xp, const double *&xd) {
        const IpmDir &dst = (mode == 2) ? s.D[1 - s.cur] : s.D[s.cur]; dx = dst.x; dy = dst.y; xp = s.v.xip; xd = s.v.xid; });
    if (rc != TLPK_OK) return rc;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        const IpmDir &acc = s.D[s.cur];
        const IpmDir &dst = (mode == 2) ? s.D[1 - s.cur] : s.D[s.cur];
        HIPCHK(h, hipSetDevice(c->device));
        const int nb = ipm_launch_newton_post(c->stream, s.v, dst, acc, mode == 2 ? 1 : 0, 0.0, s.partials[1]);   // dtau = 0: hx, hy (zeros) unused
        ipm_launch_finalize(c->stream, nb, 0, 0, 2, s.partials[1], s.d_out);
    }
    double q[IPM_SLOTS];
    if ((rc = gather(h, sh, 2, 0, 0, q)) != TLPK_OK) return rc;
    if ((rc = sync_all(h, sh)) != TLPK_OK) return rc;
    out[0] = q[0]; out[1] = q[1];
    return TLPK_OK;
}

/* MPC/step.jl:246-258, 290-296: out[0] = complementarity of the point moved by (ap, ad) along the accepted direction,
 * out[1] = xl'zl + xu'zu of the point itself */
int tlpk_mpc_gap(tlpk_handle *h, double ap, double ad, double *out) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    if (!out) return TLPK_BADARG;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        const int nb = mpc_launch_gap(c->stream, s.v, s.D[s.cur], ap, ad, s.partials[0]);
        ipm_launch_finalize(c->stream, nb, 2, 0, 0, s.partials[0], s.d_out);
    }
    double q[IPM_SLOTS];
    if (int rc = gather(h, sh, 2, 2, 0, q)) return rc;
    out[0] = q[0]; out[1] = q[1];
    return TLPK_OK;
}

/* MPC/step.jl:329-358 (compute_target!): targets of the centrality corrector from the accepted direction at the trial
 * step lengths (ap_, ad_), box [tmin, tmax]; they stay on the device for tlpk_mpc_newton(mode 2) */
int tlpk_mpc_targets(tlpk_handle *h, double ap_, double ad_, double tmin, double tmax) {
    Shards sh;
    if (int rc = ipm_shards(h, sh)) return rc;
    for (int r = 0; r < sh.n; ++r) {
        tlpk_handle *c = sh.c[r]; IpmState &s = *c->ipm;
        HIPCHK(h, hipSetDevice(c->device));
        ipm_launch_targets(c->stream, s.v, s.D[s.cur], ap_, ad_, tmin, tmax, s.partials[0]);
    }
    return TLPK_OK;
}

/* MPC/step.jl:112-123: primal side += ap * D, dual side += ad * D; out[0] = xl'zl + xu'zu of the new point */
int tlpk_mpc_advance(tlpk_handle *h, double ap, double ad, double *out) { return advance_all(h, ap, ad, out); }

}  // extern "C"
