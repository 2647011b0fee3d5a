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
};

namespace {

int ipm_ready(tlpk_handle *h) {
    if (!h) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "device-resident IPM is single-device (multi-device handles take tlpk_update / tlpk_solve)"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (!h->ipm) { h->last_error = "tlpk_ipm_load has not been called"; return TLPK_BADARG; }
    if (h->opt.nranks > 1) { h->last_error = "the device-resident IPM vectors are single-rank"; return TLPK_BADARG; }
    return TLPK_OK;
}
// copy `count` finalised scalars to the host (blocking)
int fetch(tlpk_handle *h, int count) {
    IpmState &s = *h->ipm;
    HIPCHK(h, hipMemcpyAsync(s.h_out, s.d_out, (size_t)count * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return TLPK_OK;
}

}  // namespace

void ipm_free(tlpk_handle *h) {
    if (!h || !h->ipm) return;
    if (h->ipm->h_out) hipHostFree(h->ipm->h_out);
    delete h->ipm;                  // device vectors are in h->allocs
    h->ipm = nullptr;
}

extern "C" {

static int ipm_load_impl(tlpk_handle *h, const double *b, const double *c, const double *l, const double *u);

int tlpk_ipm_load(tlpk_handle *h, const double *b, const double *c, const double *l, const double *u) {
    // a failed load leaves no half-initialised state behind: later tlpk_ipm_* calls report "not loaded", a retry is possible
    if (h && h->ipm) { h->last_error = "tlpk_ipm_load called twice on one handle"; return TLPK_BADARG; }
    const int rc = ipm_load_impl(h, b, c, l, u);
    if (rc != TLPK_OK && h) ipm_free(h);
    return rc;
}

static int ipm_load_impl(tlpk_handle *h, const double *b, const double *c, const double *l, const double *u) {
    if (!h || !b || !c || !l || !u) return TLPK_BADARG;
    if (!h->sub.empty()) { h->last_error = "device-resident IPM is single-device (multi-device handles take tlpk_update / tlpk_solve)"; return TLPK_BADARG; }
    if (!h->has_device) return TLPK_NO_DEVICE;
    if (h->opt.nranks > 1) { h->last_error = "the device-resident IPM vectors are single-rank"; return TLPK_BADARG; }
    HIPCHK(h, hipSetDevice(h->device));
    const bool k2 = h->S.system == 1;
    const i64 m = k2 ? h->S.k2_m : h->S.m, n = k2 ? h->S.k2_n : h->S.n;
    IpmState *sp = new (std::nothrow) IpmState();
    if (!sp) return TLPK_OOM;
    h->ipm = sp;
    IpmState &s = *sp;
    IpmVecs &v = s.v;
    v.m = m; v.n = n;
    int rc;
    if (!k2) { v.Ap = h->d.Ap; v.Ai = h->d.Ai; v.Ax = h->d.Ax; v.Tp = h->d.Tp; v.Tj = h->d.Tj; v.Tx = h->d.Tx; }
    else {
        // K2 handle: the analyse phase holds the incidence matrix of the augmented system (column p = entry p of A: 1 on
        // variable node j, A[i,j] on constraint node n + i, in A's column-major entry order) -- rebuild A (CSC + CSR) from
        // it for the residual / right-hand-side kernels
        const Symbolic &S = h->S;
        const i64 nnz = S.n;
        std::vector<i64> ap((size_t)n + 1, 0), tp((size_t)m + 1, 0);
        std::vector<i32> ai((size_t)nnz), tj((size_t)nnz);
        std::vector<double> ax((size_t)nnz), tx((size_t)nnz);
        for (i64 p = 0; p < nnz; ++p) {
            if (S.Ap[(size_t)p + 1] - S.Ap[(size_t)p] != 2) { h->last_error = "K2 incidence matrix: unexpected column"; return TLPK_INTERNAL; }
            const i64 q = S.Ap[(size_t)p];
            const i32 a = S.Ai[(size_t)q], b = S.Ai[(size_t)q + 1];
            const bool afirst = a < (i32)n;                       // the variable node is the smaller index
            const i32 j = afirst ? a : b, i = (afirst ? b : a) - (i32)n;
            ai[(size_t)p] = i; ax[(size_t)p] = S.Ax[(size_t)q + (afirst ? 1 : 0)];
            ++ap[(size_t)j + 1]; ++tp[(size_t)i + 1];
        }
        for (i64 j = 0; j < n; ++j) ap[(size_t)j + 1] += ap[(size_t)j];
        for (i64 i = 0; i < m; ++i) tp[(size_t)i + 1] += tp[(size_t)i];
        // entries of A come column by column (p ascending = j non-decreasing): ap is consistent with ai / ax as stored
        { std::vector<i64> cur(tp.begin(), tp.end() - 1); i64 p = 0;
          for (i64 j = 0; j < n; ++j) for (; p < ap[(size_t)j + 1]; ++p) { const i64 c = cur[(size_t)ai[(size_t)p]]++; tj[(size_t)c] = (i32)j; tx[(size_t)c] = ax[(size_t)p]; } }
        i64 *dp; i32 *di; double *dxv;
        if ((rc = dev_upload(h, &dp, ap)) != TLPK_OK) return rc; v.Ap = dp;
        if ((rc = dev_upload(h, &di, ai)) != TLPK_OK) return rc; v.Ai = di;
        if ((rc = dev_upload(h, &dxv, ax)) != TLPK_OK) return rc; v.Ax = dxv;
        if ((rc = dev_upload(h, &dp, tp)) != TLPK_OK) return rc; v.Tp = dp;
        if ((rc = dev_upload(h, &di, tj)) != TLPK_OK) return rc; v.Tj = di;
        if ((rc = dev_upload(h, &dxv, tx)) != TLPK_OK) return rc; v.Tx = dxv;
    }
    // problem data: b, c, l .* lflag, u .* uflag, flags (ipmdata.jl:46-47)
    std::vector<double> lz((size_t)n), uz((size_t)n), lf((size_t)n), uf((size_t)n);
    for (i64 j = 0; j < n; ++j) {
        const bool fl = std::isfinite(l[j]), fu = std::isfinite(u[j]);
        lf[(size_t)j] = fl ? 1.0 : 0.0; uf[(size_t)j] = fu ? 1.0 : 0.0;
        lz[(size_t)j] = fl ? l[j] : 0.0; uz[(size_t)j] = fu ? u[j] : 0.0;
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
    if (int rc = ipm_ready(h)) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    ipm_launch_init(h->stream, h->ipm->v);
    return TLPK_OK;
}

/* HSD.jl:77-128 + the quantities of HSD.jl:136-196.  out[13]:
 *  0 |rp|inf  1 |rl|inf  2 |ru|inf  3 |rd|inf  4 c'x  5 b'y  6 lz'zl  7 uz'zu  8 xl'zl + xu'zu
 *  9 |A x|inf  10 |(x - xl) lflag|inf  11 |(x + xu) uflag|inf  12 |A'y + zl lflag - zu uflag|inf */
int tlpk_ipm_residuals(tlpk_handle *h, double tau, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!out) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    const int nbc = ipm_launch_res_cols(h->stream, s.v, tau, s.partials[0]);
    const int nbr = ipm_launch_res_rows(h->stream, s.v, tau, s.partials[1]);
    ipm_launch_finalize(h->stream, nbc, 4, 6, 0, s.partials[0], s.d_out);
    ipm_launch_finalize(h->stream, nbr, 1, 2, 0, s.partials[1], s.d_out + IPM_SLOTS);
    if (int rc = fetch(h, 2 * IPM_SLOTS)) return rc;
    const double *a = s.h_out, *r = s.h_out + IPM_SLOTS;
    out[0] = r[1]; out[1] = a[4]; out[2] = a[5]; out[3] = a[6]; out[4] = a[0]; out[5] = r[0]; out[6] = a[1]; out[7] = a[2];
    out[8] = a[3]; out[9] = r[2]; out[10] = a[7]; out[11] = a[8]; out[12] = a[9];
    return TLPK_OK;
}

/* step.jl:24-51: theta_inv from the iterate, uniform regularisations, KKT.update!.  TLPK_NOT_POSDEF is
 * the PosDefException of the retry loop: call again with larger regularisations. */
int tlpk_ipm_factor(tlpk_handle *h, double regP, double regD) {
    if (int rc = ipm_ready(h)) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    ipm_launch_theta(h->stream, h->ipm->v, h->d_theta, h->d_regP, h->d_regD, regP, regD);
    return tlpk_update_device(h, h->d_theta, h->d_regP, h->d_regD);
}

/* step.jl:56-76: solve the h-system (xi_p = b, xi_d = c - th_l lz - th_u uz); hx, hy stay on the device.
 * out[0] = lz'(lz th_l) + uz'(uz th_u) - (c + th_l lz + th_u uz)'hx + b'hy   (the host adds kappa/tau + regG) */
int tlpk_ipm_hsolve(tlpk_handle *h, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!out) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    ipm_launch_hrhs(h->stream, s.v);
    int rc = tlpk_solve_device(h, s.v.hx, s.v.hy, s.v.b, s.v.hxid);
    if (rc != TLPK_OK) return rc;
    const int nb = ipm_launch_hdots(h->stream, s.v, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 2, 0, 0, s.partials[0], s.d_out);
    if ((rc = fetch(h, 2)) != TLPK_OK) return rc;
    rc = tlpk_sync(h);
    if (rc != TLPK_OK) return rc;
    out[0] = s.h_out[0] + s.h_out[1];
    return TLPK_OK;
}

/* step.jl:325-364 (first half of compute_higher_corrector): targets from the accepted direction;
 * out[0] = sum(vl), out[1] = sum(vu); the host adds the tau-kappa term and forms delta. */
int tlpk_ipm_targets(tlpk_handle *h, double a_, double mu_l, double mu_u, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!out) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    const int nb = ipm_launch_targets(h->stream, s.v, s.D[s.cur], a_, a_, mu_l, mu_u, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 2, 0, 0, s.partials[0], s.d_out);
    if (int rc = fetch(h, 2)) return rc;
    out[0] = s.h_out[0]; out[1] = s.h_out[1];
    return TLPK_OK;
}

/* step.jl:198-266 (solve_newton_system) + step.jl:294-306 (max step) for one right-hand side:
 *   mode 0 predictor, mode 1 corrector, mode 2 centrality corrector (after tlpk_ipm_targets).
 * sc[8] = { tau, kappa, h0, xi_g, xi_tk, eta, gamma*mu, delta }.
 * Modes 0 / 1 write the accepted direction, mode 2 writes the candidate (Dc = solution + accepted direction).
 * out[3] = { dtau, dkappa, largest step to the boundary over xl, xu, zl, zu (inf if none) } of the written
 * direction; for mode 2 dtau / dkappa are those of the solution alone (the host adds the accepted ones). */
int tlpk_ipm_newton(tlpk_handle *h, int mode, const double *sc, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!sc || !out || mode < 0 || mode > 2) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    const double tau = sc[0], kappa = sc[1], h0 = sc[2], xi_g = sc[3], xi_tk = sc[4], eta = sc[5], gmu = sc[6], delta = sc[7];
    const IpmDir &acc = s.D[s.cur];
    const IpmDir &dst = (mode == 2) ? s.D[1 - s.cur] : s.D[s.cur];
    const int nb = ipm_launch_newton_pre(h->stream, s.v, acc, mode, eta, gmu, delta, s.partials[0]);
    int rc = tlpk_solve_device(h, dst.x, dst.y, s.v.xip, s.v.xid);
    if (rc != TLPK_OK) return rc;
    ipm_launch_newton_dots(h->stream, s.v, dst, nb, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 6, 0, 0, s.partials[0], s.d_out);
    if ((rc = fetch(h, 6)) != TLPK_OK) return rc;
    if ((rc = tlpk_sync(h)) != TLPK_OK) return rc;
    const double *q = s.h_out;
    // step.jl:232-246
    const double xi_g_ = xi_g + xi_tk / tau - q[0] + q[1] - q[2] - q[3];
    const double dtau = (xi_g_ + q[4] - q[5]) / h0;
    const double dkappa = (xi_tk - kappa * dtau) / tau;
    const int nb2 = ipm_launch_newton_post(h->stream, s.v, dst, acc, mode == 2 ? 1 : 0, dtau, s.partials[1]);
    ipm_launch_finalize(h->stream, nb2, 0, 0, 2, s.partials[1], s.d_out);
    if ((rc = fetch(h, 2)) != TLPK_OK) return rc;
    out[0] = dtau; out[1] = dkappa; out[2] = std::fmin(s.h_out[0], s.h_out[1]);    // one step length for both sides
    return TLPK_OK;
}

/* step.jl:56-94 in ONE call: the h-system (step.jl:56-76) and the predictor's Newton system (mode 0 of tlpk_ipm_newton) are
 * independent right-hand sides against the same factor -- they share one pass over L (tlpk_solve2_device: the sweeps are bound by
 * the bytes of L).  sc[8] as for tlpk_ipm_newton, except sc[2] = regG (h0 = dot products + kappa / tau + regG is formed here).  out[4] = { dtau, dkappa, largest step to the boundary, h0 }.  Same arithmetic as tlpk_ipm_hsolve followed by
 * tlpk_ipm_newton(mode 0): bit-identical vectors and scalars. */
/* step.jl:24-94 in ONE call: tlpk_ipm_factor (theta_inv from the iterate, uniform regularisations, KKT.update!) WITHOUT the wait for its
 * status, followed by tlpk_ipm_hsolve_newton: the block-level forward sweeps of the paired solve overlap the factorisation of the root
 * (linking) front (tlpk_update_device_async).  TLPK_NOT_POSDEF is returned where tlpk_ipm_factor would have returned it -- the
 * speculative solve has only written direction / h-system buffers -- and the caller retries with larger regularisations (step.jl:35-51). */
int tlpk_ipm_factor_hsolve_newton(tlpk_handle *h, double regP, double regD, const double *sc, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!sc || !out) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    ipm_launch_theta(h->stream, h->ipm->v, h->d_theta, h->d_regP, h->d_regD, regP, regD);
    const int rc = tlpk_update_device_async(h, h->d_theta, h->d_regP, h->d_regD);
    if (rc != TLPK_OK) return rc;
    return tlpk_ipm_hsolve_newton(h, sc, out);
}

int tlpk_ipm_hsolve_newton(tlpk_handle *h, const double *sc, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!sc || !out) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    const double tau = sc[0], kappa = sc[1], regG = sc[2], xi_g = sc[3], xi_tk = sc[4], eta = sc[5], gmu = sc[6], delta = sc[7];
    const IpmDir &dst = s.D[s.cur];
    ipm_launch_hrhs(h->stream, s.v);
    const int nb = ipm_launch_newton_pre(h->stream, s.v, dst, 0, eta, gmu, delta, s.partials[0]);
    int rc = tlpk_solve2_device(h, s.v.hx, s.v.hy, s.v.b, s.v.hxid, dst.x, dst.y, s.v.xip, s.v.xid);
    if (rc != TLPK_OK) return rc;
    const int nbh = ipm_launch_hdots(h->stream, s.v, s.partials[1]);
    ipm_launch_finalize(h->stream, nbh, 2, 0, 0, s.partials[1], s.d_out + IPM_SLOTS);
    ipm_launch_newton_dots(h->stream, s.v, dst, nb, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 6, 0, 0, s.partials[0], s.d_out);
    if ((rc = fetch(h, IPM_SLOTS + 2)) != TLPK_OK) return rc;
    if ((rc = tlpk_sync(h)) != TLPK_OK) return rc;
    const double *q = s.h_out;
    const double h0 = ((q[IPM_SLOTS] + q[IPM_SLOTS + 1]) + kappa / tau) + regG;      // the association of hsd_device.py / HSD/step.jl:69-76
    const double xi_g_ = xi_g + xi_tk / tau - q[0] + q[1] - q[2] - q[3];
    const double dtau = (xi_g_ + q[4] - q[5]) / h0;
    const double dkappa = (xi_tk - kappa * dtau) / tau;
    const int nb2 = ipm_launch_newton_post(h->stream, s.v, dst, dst, 0, dtau, s.partials[1]);
    ipm_launch_finalize(h->stream, nb2, 0, 0, 2, s.partials[1], s.d_out);
    if ((rc = fetch(h, 2)) != TLPK_OK) return rc;
    out[0] = dtau; out[1] = dkappa; out[2] = std::fmin(s.h_out[0], s.h_out[1]); out[3] = h0;
    return TLPK_OK;
}

/* step.jl:112-118: the candidate of the last mode-2 call becomes the accepted direction */
int tlpk_ipm_accept(tlpk_handle *h) {
    if (int rc = ipm_ready(h)) return rc;
    h->ipm->cur = 1 - h->ipm->cur;
    return TLPK_OK;
}

/* step.jl:139-148: pt += alpha * D; out[0] = xl'zl + xu'zu of the new point (mu numerator, point.jl:45-48) */
int tlpk_ipm_advance(tlpk_handle *h, double alpha, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!out) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    const int nb = ipm_launch_advance(h->stream, s.v, s.D[s.cur], alpha, alpha, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 1, 0, 0, s.partials[0], s.d_out);
    if (int rc = fetch(h, 1)) return rc;
    out[0] = s.h_out[0];
    return TLPK_OK;
}

/* download one vector of the iterate: what = 0 x, 1 xl, 2 xu, 3 zl, 4 zu (length n), 5 y (length m) */
int tlpk_ipm_get(tlpk_handle *h, int what, double *host, int64_t len) {
    if (int rc = ipm_ready(h)) return rc;
    if (!host || what < 0 || what > 5) return TLPK_BADARG;
    const IpmVecs &v = h->ipm->v;
    const double *src[6] = {v.x, v.xl, v.xu, v.zl, v.zu, v.y};
    const int64_t need = (what == 5) ? v.m : v.n;
    if (len != need) { h->last_error = "tlpk_ipm_get: wrong length"; return TLPK_BADARG; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (need > 0) HIPCHK(h, hipMemcpy(host, src[what], (size_t)need * 8, hipMemcpyDeviceToHost));
    return TLPK_OK;
}

/* ---- Mehrotra predictor-corrector with the iterate in HBM (MPC/MPC.jl, MPC/step.jl) ------------------------
 * Shares the vectors, tlpk_ipm_load / residuals (tau = 1) / factor / accept / get with the HSD entry points. */

/* MPC.jl:353-410: starting point.  One factorisation of A A' + 1e-6 I, two solves with a zero half of the
 * right-hand side, shifts to positive coordinates, balanced products.  out[0] = xl'zl + xu'zu. */
int tlpk_mpc_start(tlpk_handle *h, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!out) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    const IpmVecs &v = s.v;
    mpc_launch_fill(h->stream, v, h->d_theta, h->d_regP, h->d_regD);
    int rc = tlpk_update_device(h, h->d_theta, h->d_regP, h->d_regD);
    if (rc != TLPK_OK) return rc;
    if ((rc = tlpk_solve_device(h, s.D[0].x, v.y, v.xip, v.c)) != TLPK_OK) return rc;          // y  (xip == 0 here)
    if ((rc = tlpk_solve_device(h, v.x, s.D[0].y, v.b, v.xid)) != TLPK_OK) return rc;          // x  (xid == 0 here)
    int nb = mpc_launch_start(h->stream, v, 1, 0.0, 0.0, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 0, 0, 2, s.partials[0], s.d_out);
    if ((rc = fetch(h, 2)) != TLPK_OK) return rc;
    if ((rc = tlpk_sync(h)) != TLPK_OK) return rc;
    const double dxs = 1.0 + std::fmax(0.0, std::fmax(-1.5 * s.h_out[0], -1.5 * s.h_out[1]));
    nb = mpc_launch_start(h->stream, v, 2, dxs, 0.0, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 0, 0, 2, s.partials[0], s.d_out);
    if ((rc = fetch(h, 2)) != TLPK_OK) return rc;
    const double dzs = 1.0 + std::fmax(0.0, std::fmax(-1.5 * s.h_out[0], -1.5 * s.h_out[1]));
    nb = mpc_launch_start(h->stream, v, 3, dzs, 0.0, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 3, 0, 0, s.partials[0], s.d_out);
    if ((rc = fetch(h, 3)) != TLPK_OK) return rc;
    const double mu = s.h_out[0], ddx = mu / (2.0 * s.h_out[1]), ddz = mu / (2.0 * s.h_out[2]);
    nb = mpc_launch_start(h->stream, v, 4, ddx, ddz, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 1, 0, 0, s.partials[0], s.d_out);
    if ((rc = fetch(h, 1)) != TLPK_OK) return rc;
    out[0] = s.h_out[0];
    s.cur = 0;
    return TLPK_OK;
}

/* MPC/step.jl:164-217 (solve_newton_system + max_step_length_pd) for one right-hand side:
 *   mode 0 predictor (xi = residuals, complementarity -x z), mode 1 corrector (same residuals, sigma mu - x z - dx dz
 *   with the predictor direction, which it overwrites; gmu = sigma mu), mode 2 centrality corrector (zero residuals,
 *   the targets of tlpk_mpc_targets; writes the candidate = solution + accepted direction).
 * out[2] = largest primal step (over xl, xu) and largest dual step (over zl, zu) to the boundary, inf if none. */
int tlpk_mpc_newton(tlpk_handle *h, int mode, double gmu, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!out || mode < 0 || mode > 2) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    const IpmDir &acc = s.D[s.cur];
    const IpmDir &dst = (mode == 2) ? s.D[1 - s.cur] : s.D[s.cur];
    ipm_launch_newton_pre(h->stream, s.v, acc, mode, 1.0, gmu, 0.0, s.partials[0]);
    int rc = tlpk_solve_device(h, dst.x, dst.y, s.v.xip, s.v.xid);
    if (rc != TLPK_OK) return rc;
    const int nb = ipm_launch_newton_post(h->stream, s.v, dst, acc, mode == 2 ? 1 : 0, 0.0, s.partials[1]);   // dtau = 0: hx, hy (zeros) unused
    ipm_launch_finalize(h->stream, nb, 0, 0, 2, s.partials[1], s.d_out);
    if ((rc = fetch(h, 2)) != TLPK_OK) return rc;
    if ((rc = tlpk_sync(h)) != TLPK_OK) return rc;
    out[0] = s.h_out[0]; out[1] = s.h_out[1];
    return TLPK_OK;
}

/* MPC/step.jl:246-258, 290-296: out[0] = complementarity of the point moved by (ap, ad) along the accepted direction,
 * out[1] = xl'zl + xu'zu of the point itself */
int tlpk_mpc_gap(tlpk_handle *h, double ap, double ad, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!out) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    const int nb = mpc_launch_gap(h->stream, s.v, s.D[s.cur], ap, ad, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 2, 0, 0, s.partials[0], s.d_out);
    if (int rc = fetch(h, 2)) return rc;
    out[0] = s.h_out[0]; out[1] = s.h_out[1];
    return TLPK_OK;
}

/* MPC/step.jl:329-358 (compute_target!): targets of the centrality corrector from the accepted direction at the trial
 * step lengths (ap_, ad_), box [tmin, tmax]; they stay on the device for tlpk_mpc_newton(mode 2) */
int tlpk_mpc_targets(tlpk_handle *h, double ap_, double ad_, double tmin, double tmax) {
    if (int rc = ipm_ready(h)) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    ipm_launch_targets(h->stream, s.v, s.D[s.cur], ap_, ad_, tmin, tmax, s.partials[0]);
    return TLPK_OK;
}

/* MPC/step.jl:112-123: primal side += ap * D, dual side += ad * D; out[0] = xl'zl + xu'zu of the new point */
int tlpk_mpc_advance(tlpk_handle *h, double ap, double ad, double *out) {
    if (int rc = ipm_ready(h)) return rc;
    if (!out) return TLPK_BADARG;
    HIPCHK(h, hipSetDevice(h->device));
    IpmState &s = *h->ipm;
    const int nb = ipm_launch_advance(h->stream, s.v, s.D[s.cur], ap, ad, s.partials[0]);
    ipm_launch_finalize(h->stream, nb, 1, 0, 0, s.partials[0], s.d_out);
    if (int rc = fetch(h, 1)) return rc;
    out[0] = s.h_out[0];
    return TLPK_OK;
}

}  // extern "C"
