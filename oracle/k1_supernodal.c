/*
 * k1_supernodal.c -- CPU comparator for the normal-equations (K1) KKT path: supernodal
 * multifrontal Cholesky on dense BLAS-3 kernels, threaded over the host cores.
 *
 * THIS IS TEST / BENCHMARK INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/ and bench.py's
 * cpu_baseline leg may load this library (oracle/libk1sn.so); libtlpk.so never links, loads or
 * calls it.
 *
 * Why it exists: the reference's K1 backend (/root/reference/src/KKT/Cholmod/spd.jl:22-70) hands
 * `A*D*A' + Rd` to SuiteSparse CHOLMOD [ext], whose numeric phase is a SUPERNODAL factorisation on
 * dense BLAS-3 (dpotrf / dtrsm / dsyrk per supernode), run with the BLAS threads Tulip sets
 * (/root/reference/src/model.jl:73, `Threads`).  Neither Julia nor CHOLMOD exists in this image
 * (BASELINE.md section 2), and the parity oracle (k1_oracle.c) is a 1-core simplicial code -- a
 * strawman as a speed baseline.  This file is the CHOLMOD-class CPU path restated: same
 * semantics as spd.jl (D = 1/(theta+regP); S = A*D*A' + diag(regD); P S P' = L L'; no pivot
 * perturbation; xi = xi_p + A*(D.*xi_d); dy = S\xi; dx = D.*(A'dy - xi_d)), supernodal
 * multifrontal numeric phase (Duff & Reid 1983; Liu 1992; Chen, Davis, Hager, Rajamanickam, ACM TOMS
 * 35(3) 2008 section 3) on the OpenBLAS that ships with SciPy (dlopen'ed: `scipy_dpotrf_`,
 * `scipy_dtrsm_`, `scipy_dsyrk_`, `scipy_dgemv_`, `scipy_dtrsv_`), tree-level parallel with OpenMP; the triangular
 * solves additionally split every large front over the whole team (round 4: see solve_level)
 * (levels with many fronts: one front per thread, BLAS single-threaded; levels with few fronts:
 * one front at a time, BLAS on all threads).
 *
 * The symbolic structure (ordering, supernodes/fronts, relative indices, assembly lists of
 * A*D*A') is NOT recomputed here: the caller passes the arrays of an analyse-only libtlpk handle
 * (tlpk_symbolic_get; panels are f x ns column-major with leading dimension lda >= f), so that the CPU and GPU paths factorise the same permuted matrix with the
 * same supernode partition => same nnz(L), same flops, and the factor panels can be compared entry
 * by entry at full benchmark size (tests/test_gpu_parity.py).  The numeric code is independent of
 * the HIP kernels (LAPACK-style unblocked-by-us calls vs hand-written MFMA tiles).
 *
 * Pinned by tests/test_oracle.py: against k1_oracle.c (L entrywise, dx, dy), the reference
 * fixture and the golden vectors.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;
typedef int blasint;

typedef void (*dpotrf_t)(const char *, const blasint *, double *, const blasint *, blasint *);
typedef void (*dtrsm_t)(const char *, const char *, const char *, const char *, const blasint *, const blasint *,
                        const double *, const double *, const blasint *, double *, const blasint *);
typedef void (*dsyrk_t)(const char *, const char *, const blasint *, const blasint *, const double *, const double *,
                        const blasint *, const double *, double *, const blasint *);
typedef void (*dgemv_t)(const char *, const blasint *, const blasint *, const double *, const double *, const blasint *,
                        const double *, const blasint *, const double *, double *, const blasint *);
typedef void (*dtrsv_t)(const char *, const char *, const char *, const blasint *, const double *, const blasint *,
                        double *, const blasint *);
typedef void (*setnt_t)(int);
typedef int (*getnt_t)(void);

typedef struct k1sn {
    i64 m, n, nnzA, nf, nnzS, lval_len, nlevels;
    /* A: CSC, 0-based */
    i64 *Ap, *Ai; double *Ax;
    i64 *Tp, *Tj; double *Tx;                     /* the same matrix by rows (solve: xi = xi_p + A (D .* xi_d), one row per thread, fixed order) */
    i64 *perm;                                    /* perm[new] = old */
    /* fronts */
    i64 *f, *ns, *col0, *loff, *rowoff, *reloff, *child_ptr, *nchild, *depth, *lda;
    i64 *rowidx, *rel, *children;
    i64 *level_ptr, *level_fronts;                /* fronts by depth, heaviest first inside a level */
    i64 *ucoff; i64 uc_len;
    /* assembly lists */
    i64 *s_target, *s_diag_row, *pair_ptr, *pair_j; double *pair_w;
    /* numeric */
    double *theta, *regP, *regD, *D, *Lval, **U, *uc, *xw;
    int factored; i64 fail_col;
    int nthreads;
    double t_assemble, t_factor, t_solve;
    void *blas;
    dpotrf_t dpotrf; dtrsm_t dtrsm; dsyrk_t dsyrk; dgemv_t dgemv; dtrsv_t dtrsv; setnt_t setnt;
} k1sn;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* CPUs granted by the CFS bandwidth controller (rounded up), 0 = no limit found */
static int cgroup_cpu_quota(void) {
    double quota = -1.0, period = -1.0;
    FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (fp) {
        char q[64];
        if (fscanf(fp, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
        fclose(fp);
    } else {
        FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fpp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fpp && fscanf(fq, "%lf", &quota) == 1 && fscanf(fpp, "%lf", &period) == 1) { /* read */ } else quota = -1.0;
        if (fq) fclose(fq);
        if (fpp) fclose(fpp);
    }
    if (quota <= 0.0 || period <= 0.0) return 0;
    const int n = (int)((quota + period - 1.0) / period);
    return n > 0 ? n : 1;
}

static i64 *dup64(const i64 *src, i64 n) {
    i64 *p = (i64 *)malloc((size_t)(n > 0 ? n : 1) * sizeof(i64));
    if (p && n > 0) memcpy(p, src, (size_t)n * sizeof(i64));
    return p;
}
static double *dupd(const double *src, i64 n) {
    double *p = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    if (p && n > 0) memcpy(p, src, (size_t)n * sizeof(double));
    return p;
}

void k1sn_free(k1sn *h) {
    if (!h) return;
    free(h->Ap); free(h->Ai); free(h->Ax); free(h->perm); free(h->Tp); free(h->Tj); free(h->Tx);
    free(h->f); free(h->ns); free(h->col0); free(h->loff); free(h->rowoff); free(h->reloff);
    free(h->child_ptr); free(h->nchild); free(h->depth); free(h->lda); free(h->rowidx); free(h->rel); free(h->children);
    free(h->level_ptr); free(h->level_fronts); free(h->ucoff);
    free(h->s_target); free(h->s_diag_row); free(h->pair_ptr); free(h->pair_j); free(h->pair_w);
    free(h->theta); free(h->regP); free(h->regD); free(h->D); free(h->Lval); free(h->uc); free(h->xw);
    if (h->U) { for (i64 s = 0; s < h->nf; ++s) free(h->U[s]); free(h->U); }
    if (h->blas) dlclose(h->blas);
    free(h);
}

/* sort helper: fronts of a level by decreasing work */
typedef struct { double w; i64 s; } wk_t;
static int cmp_wk(const void *a, const void *b) { const double x = ((const wk_t *)a)->w, y = ((const wk_t *)b)->w; return (x < y) - (x > y); }

/* All index arrays are int64, 0-based, exactly as tlpk_symbolic_get returns them.
 * returns 0 ok, 2 bad argument / BLAS not found, 3 out of memory */
int k1sn_create(k1sn **out, const char *blas_path, i64 m, i64 n, const i64 *Ap, const i64 *Ai, const double *Ax,
                const i64 *perm, i64 nf, const i64 *f, const i64 *ns, const i64 *col0, const i64 *loff,
                const i64 *rowoff, const i64 *reloff, const i64 *child_ptr, const i64 *nchild, const i64 *depth, const i64 *lda,
                i64 n_rowidx, const i64 *rowidx, i64 n_rel, const i64 *rel, i64 n_children, const i64 *children,
                i64 nnzS, const i64 *s_target, const i64 *s_diag_row, const i64 *pair_ptr, const i64 *pair_j,
                const double *pair_w, i64 lval_len, int nthreads) {
    if (!out) return 2;
    *out = NULL;
    k1sn *h = (k1sn *)calloc(1, sizeof(k1sn));
    if (!h) return 3;
    h->blas = dlopen(blas_path, RTLD_NOW | RTLD_LOCAL);
    if (!h->blas) { fprintf(stderr, "k1sn: dlopen(%s): %s\n", blas_path ? blas_path : "(null)", dlerror()); free(h); return 2; }
    *(void **)&h->dpotrf = dlsym(h->blas, "scipy_dpotrf_"); *(void **)&h->dtrsm = dlsym(h->blas, "scipy_dtrsm_");
    *(void **)&h->dsyrk = dlsym(h->blas, "scipy_dsyrk_"); *(void **)&h->dgemv = dlsym(h->blas, "scipy_dgemv_");
    *(void **)&h->dtrsv = dlsym(h->blas, "scipy_dtrsv_"); *(void **)&h->setnt = dlsym(h->blas, "scipy_openblas_set_num_threads");
    if (!h->dpotrf || !h->dtrsm || !h->dsyrk || !h->dgemv || !h->dtrsv || !h->setnt) { k1sn_free(h); return 2; }
    h->m = m; h->n = n; h->nnzA = Ap[n]; h->nf = nf; h->nnzS = nnzS; h->lval_len = lval_len;
#ifdef _OPENMP
    h->nthreads = nthreads > 0 ? nthreads : omp_get_max_threads();
    if (nthreads == 0) {
        /* "all cores" means the cores this process may USE: inside a container the CFS quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us) is usually
         * far below the number of online CPUs.  Measured on the GPU box (256 hardware threads online, quota 16 CPUs): 64 threads are throttled into
         * each other -- factorisation 2.4 s and 0.40 s per solve of config C4, against 1.4 s and 0.11 s with 16 threads. */
        const int q = cgroup_cpu_quota();
        if (q > 0 && q < h->nthreads) h->nthreads = q;
    }
#else
    h->nthreads = 1; (void)nthreads;
#endif
    {
        /* The OpenBLAS inside the SciPy wheel is compiled for a fixed maximum number of threads (64 in this
         * image): more concurrent callers than that corrupt its buffer table (observed on a 256-thread host:
         * "precompiled NUM_THREADS exceeded" followed by heap corruption).  At load time
         * openblas_get_num_threads() = min(compiled maximum, cores): never run more threads than that. */
        getnt_t getnt; *(void **)&getnt = dlsym(h->blas, "scipy_openblas_get_num_threads");
        const int cap = getnt ? getnt() : 1;
        if (cap >= 1 && h->nthreads > cap) h->nthreads = cap;
    }
    h->Ap = dup64(Ap, n + 1); h->Ai = dup64(Ai, h->nnzA); h->Ax = dupd(Ax, h->nnzA); h->perm = dup64(perm, m);
    h->f = dup64(f, nf); h->ns = dup64(ns, nf); h->col0 = dup64(col0, nf); h->loff = dup64(loff, nf);
    h->rowoff = dup64(rowoff, nf); h->reloff = dup64(reloff, nf); h->child_ptr = dup64(child_ptr, nf);
    h->nchild = dup64(nchild, nf); h->depth = dup64(depth, nf); h->lda = dup64(lda, nf);
    h->rowidx = dup64(rowidx, n_rowidx); h->rel = dup64(rel, n_rel); h->children = dup64(children, n_children);
    h->s_target = dup64(s_target, nnzS); h->s_diag_row = dup64(s_diag_row, nnzS); h->pair_ptr = dup64(pair_ptr, nnzS + 1);
    h->pair_j = dup64(pair_j, pair_ptr[nnzS]); h->pair_w = dupd(pair_w, pair_ptr[nnzS]);
    h->theta = (double *)calloc((size_t)(n > 0 ? n : 1), 8); h->regP = (double *)calloc((size_t)(n > 0 ? n : 1), 8);
    h->regD = (double *)calloc((size_t)(m > 0 ? m : 1), 8); h->D = (double *)calloc((size_t)(n > 0 ? n : 1), 8);
    h->Lval = (double *)malloc((size_t)(lval_len > 0 ? lval_len : 1) * 8);
    h->U = (double **)calloc((size_t)(nf > 0 ? nf : 1), sizeof(double *));
    h->xw = (double *)calloc((size_t)(m > 0 ? m : 1), 8);
    h->ucoff = (i64 *)malloc((size_t)(nf > 0 ? nf : 1) * sizeof(i64));
    if (!h->Ap || !h->Ai || !h->Ax || !h->perm || !h->f || !h->ns || !h->col0 || !h->loff || !h->rowoff || !h->reloff ||
        !h->child_ptr || !h->nchild || !h->depth || !h->lda || !h->rowidx || !h->rel || !h->children || !h->s_target || !h->s_diag_row ||
        !h->pair_ptr || !h->pair_j || !h->pair_w || !h->theta || !h->regP || !h->regD || !h->D || !h->Lval || !h->U || !h->xw || !h->ucoff) {
        k1sn_free(h); return 3;
    }
    i64 maxd = -1;
    for (i64 s = 0; s < nf; ++s) { h->ucoff[s] = h->uc_len; h->uc_len += f[s] - ns[s]; if (depth[s] > maxd) maxd = depth[s]; }
    h->uc = (double *)calloc((size_t)(h->uc_len > 0 ? h->uc_len : 1), 8);
    h->nlevels = maxd + 1;
    h->level_ptr = (i64 *)calloc((size_t)(h->nlevels + 2), sizeof(i64));
    h->level_fronts = (i64 *)malloc((size_t)(nf > 0 ? nf : 1) * sizeof(i64));
    wk_t *wk = (wk_t *)malloc((size_t)(nf > 0 ? nf : 1) * sizeof(wk_t));
    if (!h->uc || !h->level_ptr || !h->level_fronts || !wk) { free(wk); k1sn_free(h); return 3; }
    for (i64 s = 0; s < nf; ++s) h->level_ptr[depth[s] + 1]++;
    for (i64 d = 0; d < h->nlevels; ++d) h->level_ptr[d + 1] += h->level_ptr[d];
    {
        i64 *fill = dup64(h->level_ptr, h->nlevels + 1);
        if (!fill) { free(wk); k1sn_free(h); return 3; }
        for (i64 s = 0; s < nf; ++s) h->level_fronts[fill[depth[s]]++] = s;
        free(fill);
    }
    for (i64 d = 0; d < h->nlevels; ++d) {
        const i64 a = h->level_ptr[d], b = h->level_ptr[d + 1];
        for (i64 t = a; t < b; ++t) { const i64 s = h->level_fronts[t]; wk[t - a].s = s; wk[t - a].w = (double)f[s] * (double)ns[s] * (double)ns[s]; }
        qsort(wk, (size_t)(b - a), sizeof(wk_t), cmp_wk);
        for (i64 t = a; t < b; ++t) h->level_fronts[t] = wk[t - a].s;
    }
    free(wk);
    {
        /* Small problems (a Netlib-size LP: ~1e7 factor flops) run on ONE thread: forking an OpenMP team and re-sizing the
         * BLAS thread pool per tree level costs milliseconds per call, 100x the arithmetic (CHOLMOD itself stays serial
         * below its own thresholds).  nthreads_arg < 0 keeps the requested team whatever the size. */
        double work = 0.0;
        for (i64 s = 0; s < nf; ++s) work += (double)f[s] * (double)f[s] * (double)ns[s];
        if (nthreads >= 0 && work < 2.0e8) h->nthreads = 1;
        h->setnt(h->nthreads);
    }
    {   /* rows of A (column indices ascending inside a row: the order of the column loop it replaces) */
        h->Tp = (i64 *)calloc((size_t)(m + 2), sizeof(i64)); h->Tj = (i64 *)malloc((size_t)(h->nnzA > 0 ? h->nnzA : 1) * sizeof(i64));
        h->Tx = (double *)malloc((size_t)(h->nnzA > 0 ? h->nnzA : 1) * 8);
        if (!h->Tp || !h->Tj || !h->Tx) { k1sn_free(h); return 3; }
        for (i64 p = 0; p < h->nnzA; ++p) h->Tp[Ai[p] + 2]++;
        for (i64 i = 0; i < m; ++i) h->Tp[i + 2] += h->Tp[i + 1];
        for (i64 j = 0; j < n; ++j)
            for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) { const i64 q = h->Tp[Ai[p] + 1]++; h->Tj[q] = j; h->Tx[q] = Ax[p]; }
    }
    h->fail_col = -1;
    *out = h;
    return 0;
}

#define SMALL_PANEL 2048       /* fronts with at most this many panel entries never call the BLAS (see fwd_front) */

static i64 small_factor_limit(void) { static i64 v = -1; if (v < 0) { const char *e = getenv("K1SN_SMALL_FACTOR"); v = e ? atoll(e) : SMALL_PANEL; } return v; }

/* one front: extend-add of the children's update matrices, dense partial factorisation */
static void factor_front(k1sn *h, i64 s, i64 *fail) {
    const i64 f = h->f[s], ns = h->ns[s], rs = f - ns, ld = h->lda[s];       /* panel: f x ns, leading dimension ld >= f */
    double *P = h->Lval + h->loff[s];
    double *Us = NULL;
    if (rs > 0) { Us = (double *)calloc((size_t)(rs * rs), 8); h->U[s] = Us; if (!Us) { *fail = -2; return; } }
    for (i64 ci = 0; ci < h->nchild[s]; ++ci) {
        const i64 c = h->children[h->child_ptr[s] + ci];
        const i64 rsc = h->f[c] - h->ns[c];
        const i64 *relc = h->rel + h->reloff[c];
        const double *Uc = h->U[c];
        if (rsc == 0 || !Uc) continue;
        for (i64 q = 0; q < rsc; ++q) {
            const i64 tc = relc[q];
            const double *src = Uc + q * rsc;
            if (tc < ns) { double *dst = P + tc * ld; for (i64 r = q; r < rsc; ++r) dst[relc[r]] += src[r]; }
            else { double *dst = Us + (tc - ns) * rs - ns; for (i64 r = q; r < rsc; ++r) dst[relc[r]] += src[r]; }
        }
        free(h->U[c]); h->U[c] = NULL;
    }
    blasint info = 0, bn = (blasint)ns, bf = (blasint)ld, brs = (blasint)rs;
    const int small = f * ns <= small_factor_limit();
    if (small) {
        /* right-looking column Cholesky of the f x ns panel, then U -= L21 L21' (lower triangle): see fwd_front */
        for (i64 j = 0; j < ns && info == 0; ++j) {
            double *Pj = P + j * ld;
            const double d = Pj[j];
            if (!(d > 0.0)) { info = (blasint)(j + 1); break; }
            const double r = sqrt(d);
            Pj[j] = r;
            for (i64 i = j + 1; i < f; ++i) Pj[i] /= r;
            for (i64 k = j + 1; k < ns; ++k) {
                double *Pk = P + k * ld;
                const double ljk = Pj[k];
                for (i64 i = k; i < f; ++i) Pk[i] -= Pj[i] * ljk;
            }
        }
        if (info == 0)
            for (i64 j = 0; j < ns; ++j) {
                const double *Pj = P + j * ld + ns;
                for (i64 c = 0; c < rs; ++c) { const double l = Pj[c]; double *Uc = Us + c * rs; for (i64 r = c; r < rs; ++r) Uc[r] -= Pj[r] * l; }
            }
    } else
        h->dpotrf("L", &bn, P, &bf, &info);
    if (info != 0) {                                       /* not positive definite: spd.jl:46-47 */
        const i64 col = h->col0[s] + (info > 0 ? info - 1 : 0);
#pragma omp critical(k1sn_fail)
        { if (*fail == -1 || col < *fail) *fail = col; }
        return;
    }
    if (rs > 0 && !small) {
        const double one = 1.0, mone = -1.0;
        h->dtrsm("R", "L", "T", "N", &brs, &bn, &one, P, &bf, P + ns, &bf);
        h->dsyrk("L", "N", &brs, &bn, &mone, P + ns, &bf, &one, Us, &brs);
    }
}

/* levels with at least nthreads/2 fronts: one front per thread (BLAS single-threaded);
 * others: one front at a time with the BLAS on all threads */
#define FOR_LEVEL_FRONTS(h, d, BODY)                                                              \
    do {                                                                                          \
        const i64 a_ = (h)->level_ptr[d], b_ = (h)->level_ptr[(d) + 1];                          \
        if ((h)->nthreads <= 1) {                                                                 \
            for (i64 t_ = a_; t_ < b_; ++t_) { const i64 s = (h)->level_fronts[t_]; BODY; }       \
        } else if (2 * (b_ - a_) >= (i64)(h)->nthreads) {                                         \
            (h)->setnt(1);                                                                        \
            _Pragma("omp parallel for schedule(dynamic, 1) num_threads((h)->nthreads)")           \
            for (i64 t_ = a_; t_ < b_; ++t_) { const i64 s = (h)->level_fronts[t_]; BODY; }       \
        } else {                                                                                  \
            (h)->setnt((h)->nthreads);                                                            \
            for (i64 t_ = a_; t_ < b_; ++t_) { const i64 s = (h)->level_fronts[t_]; BODY; }       \
        }                                                                                         \
    } while (0)

/* returns 0 ok, 1 not positive definite (k1sn_fail_col), 3 out of memory */
int k1sn_update(k1sn *h, const double *theta, const double *regP, const double *regD) {
    if (!h || !theta || !regP || !regD) return 2;
    h->factored = 0; h->fail_col = -1;
    memcpy(h->theta, theta, (size_t)h->n * 8); memcpy(h->regP, regP, (size_t)h->n * 8); memcpy(h->regD, regD, (size_t)h->m * 8);   /* spd.jl:36-38 */
    const double t0 = now_s();
#pragma omp parallel for schedule(static) num_threads(h->nthreads)
    for (i64 j = 0; j < h->n; ++j) h->D[j] = 1.0 / (h->theta[j] + h->regP[j]);                  /* spd.jl:42 */
    {   /* zero-fill on all threads (a single-threaded memset of an 11 GB factor costs seconds and puts every page on one NUMA node) */
        const i64 chunk = (i64)1 << 20, nchunk = (h->lval_len + chunk - 1) / chunk;
#pragma omp parallel for schedule(static) num_threads(h->nthreads)
        for (i64 q = 0; q < nchunk; ++q) {
            const i64 a = q * chunk, b = (a + chunk < h->lval_len) ? a + chunk : h->lval_len;
            memset(h->Lval + a, 0, (size_t)(b - a) * 8);
        }
    }
#pragma omp parallel for schedule(static, 4096) num_threads(h->nthreads)
    for (i64 e = 0; e < h->nnzS; ++e) {                                                          /* spd.jl:43, gathered into the panels */
        double v = 0.0;
        for (i64 p = h->pair_ptr[e]; p < h->pair_ptr[e + 1]; ++p) v += h->pair_w[p] * h->D[h->pair_j[p]];
        if (h->s_diag_row[e] >= 0) v += h->regD[h->s_diag_row[e]];
        h->Lval[h->s_target[e]] = v;
    }
    const double t1 = now_s();
    i64 fail = -1;
    for (i64 d = h->nlevels - 1; d >= 0 && fail == -1; --d) FOR_LEVEL_FRONTS(h, d, factor_front(h, s, &fail));
    for (i64 s = 0; s < h->nf; ++s) { free(h->U[s]); h->U[s] = NULL; }
    if (h->nthreads > 1) h->setnt(h->nthreads);
    h->t_assemble = t1 - t0; h->t_factor = now_s() - t1;
    if (fail == -2) return 3;
    if (fail >= 0) { h->fail_col = fail; return 1; }
    h->factored = 1;
    return 0;
}

/* Fronts with a panel of at most SMALL_PANEL entries (most fronts of a sparse LP: 10^5 of them per tree level on the north-star LP) are
 * handled by plain loops.  OpenBLAS's interface routines take a process-wide lock for their work buffer on EVERY call: 16 threads
 * calling dtrsv / dgemv (dpotrf / dtrsm / dsyrk) on 20 x 3 panels spent ~20 us per front queueing for it -- 0.35 s per sweep of one tree level
 * (measured, tools/cpu_solve_probe.py), 90 % of a solve. */

static void gather_children(k1sn *h, i64 s);

static void fwd_front(k1sn *h, i64 s) {
    const i64 f = h->f[s], ns = h->ns[s], rs = f - ns, ld = h->lda[s];
    const double *P = h->Lval + h->loff[s];
    double *x = h->xw + h->col0[s], *ucs = h->uc + h->ucoff[s];
    gather_children(h, s);
    if (f * ns <= SMALL_PANEL) {
        for (i64 j = 0; j < ns; ++j) {
            const double *Pj = P + j * ld;
            const double xj = x[j] / Pj[j];
            x[j] = xj;
            for (i64 i = j + 1; i < ns; ++i) x[i] -= Pj[i] * xj;
            for (i64 r = 0; r < rs; ++r) ucs[r] -= Pj[ns + r] * xj;
        }
        return;
    }
    const blasint bn = (blasint)ns, bf = (blasint)ld, brs = (blasint)rs, inc = 1;
    h->dtrsv("L", "N", "N", &bn, P, &bf, x, &inc);
    if (rs > 0) { const double mone = -1.0, one = 1.0; h->dgemv("N", &brs, &bn, &mone, P + ns, &bf, x, &inc, &one, ucs, &inc); }
}
static void bwd_front(k1sn *h, i64 s) {
    const i64 f = h->f[s], ns = h->ns[s], rs = f - ns, ld = h->lda[s];
    const double *P = h->Lval + h->loff[s];
    double *x = h->xw + h->col0[s], *xb = h->uc + h->ucoff[s];     /* the contribution vector is free again: reuse it */
    const i64 *rows = h->rowidx + h->rowoff[s] + ns;
    for (i64 r = 0; r < rs; ++r) xb[r] = h->xw[rows[r]];
    if (f * ns <= SMALL_PANEL) {
        for (i64 j = ns - 1; j >= 0; --j) {
            const double *Pj = P + j * ld;
            double v = x[j];
            for (i64 r = 0; r < rs; ++r) v -= Pj[ns + r] * xb[r];
            for (i64 i = j + 1; i < ns; ++i) v -= Pj[i] * x[i];
            x[j] = v / Pj[j];
        }
        return;
    }
    const blasint bn = (blasint)ns, bf = (blasint)ld, brs = (blasint)rs, inc = 1;
    if (rs > 0) {
        const double mone = -1.0, one = 1.0;
        h->dgemv("T", &brs, &bn, &mone, P + ns, &bf, xb, &inc, &one, x, &inc);
    }
    h->dtrsv("L", "T", "N", &bn, P, &bf, x, &inc);
}

/* ---- threaded triangular solves ------------------------------------------------------------------------------
 * A level of the tree is solved by ONE OpenMP team (BLAS calls single-threaded inside it):
 *   - fronts with a large panel (>= SOLVE_BIG entries) one after the other, each split over the whole team: the pivot block in
 *     SOLVE_KB-wide steps (dtrsv on the diagonal block by one thread, the dgemv of the rows below / columns before it cut into
 *     one slice per thread) -- a 100-MB panel streams at the bandwidth of all cores instead of one;
 *   - the other fronts one per thread, handed out in chunks (a level of 10^5 tiny fronts must not pay one atomic per front).
 * Summation order per entry is fixed (slices partition rows / columns, never a sum): results do not depend on the thread count. */
#define SOLVE_BIG ((i64)1 << 20)
#define SOLVE_KB  256
#define SOLVE_MIN_SLICE 512

static void gather_children(k1sn *h, i64 s) {
    const i64 ns = h->ns[s], rs = h->f[s] - ns;
    double *x = h->xw + h->col0[s], *ucs = h->uc + h->ucoff[s];
    for (i64 r = 0; r < rs; ++r) ucs[r] = 0.0;
    for (i64 ci = 0; ci < h->nchild[s]; ++ci) {
        const i64 c = h->children[h->child_ptr[s] + ci];
        const i64 rsc = h->f[c] - h->ns[c];
        const i64 *relc = h->rel + h->reloff[c];
        const double *ucc = h->uc + h->ucoff[c];
        for (i64 r = 0; r < rsc; ++r) { const i64 pos = relc[r]; if (pos < ns) x[pos] += ucc[r]; else ucs[pos - ns] += ucc[r]; }
    }
}

/* slice [a, b) number t of T of the range [lo, hi), boundaries on multiples of 8 entries (cache lines) */
static void slice_of(i64 lo, i64 hi, int t, int T, i64 *a, i64 *b) {
    const i64 len = hi - lo;
    if (len < (i64)T * SOLVE_MIN_SLICE) { T = (int)(len / SOLVE_MIN_SLICE); if (T < 1) T = 1; }     /* thin slices stream badly: fewer, longer ones */
    if (t >= T) { *a = *b = hi; return; }
    const i64 per = ((len + T - 1) / T + 7) & ~(i64)7;
    *a = lo + (i64)t * per; *b = *a + per;
    if (*a > hi) *a = hi;
    if (*b > hi) *b = hi;
}

/* called by every thread of the team (t of T); contains barriers */
static void fwd_front_team(k1sn *h, i64 s, int t, int T) {
    const i64 f = h->f[s], ns = h->ns[s], ld = h->lda[s];
    const double *P = h->Lval + h->loff[s];
    double *x = h->xw + h->col0[s], *ucs = h->uc + h->ucoff[s];
    const blasint inc = 1, bf = (blasint)ld;
    const double mone = -1.0, one = 1.0;
    if (t == 0) gather_children(h, s);
#pragma omp barrier
    for (i64 k = 0; k < ns; k += SOLVE_KB) {
        const i64 kb = (ns - k < SOLVE_KB) ? ns - k : SOLVE_KB;
        const blasint bkb = (blasint)kb;
        if (t == 0) h->dtrsv("L", "N", "N", &bkb, P + k + k * ld, &bf, x + k, &inc);
#pragma omp barrier
        i64 a, b;
        slice_of(k + kb, f, t, T, &a, &b);                       /* rows below the step, pivot rows and update rows alike */
        if (b > a) {
            /* rows [a, b) may straddle ns: the pivot part updates x, the rest the contribution vector */
            const i64 a1 = a, b1 = (b < ns) ? b : ns, a2 = (a > ns) ? a : ns, b2 = b;
            if (b1 > a1) { const blasint br = (blasint)(b1 - a1); h->dgemv("N", &br, &bkb, &mone, P + a1 + k * ld, &bf, x + k, &inc, &one, x + a1, &inc); }
            if (b2 > a2) { const blasint br = (blasint)(b2 - a2); h->dgemv("N", &br, &bkb, &mone, P + a2 + k * ld, &bf, x + k, &inc, &one, ucs + (a2 - ns), &inc); }
        }
#pragma omp barrier
    }
}
static void bwd_front_team(k1sn *h, i64 s, int t, int T) {
    const i64 f = h->f[s], ns = h->ns[s], rs = f - ns, ld = h->lda[s];
    const double *P = h->Lval + h->loff[s];
    double *x = h->xw + h->col0[s], *xb = h->uc + h->ucoff[s];
    const i64 *rows = h->rowidx + h->rowoff[s] + ns;
    const blasint inc = 1, bf = (blasint)ld;
    const double mone = -1.0, one = 1.0;
    { i64 a, b; slice_of(0, rs, t, T, &a, &b); for (i64 r = a; r < b; ++r) xb[r] = h->xw[rows[r]]; }
#pragma omp barrier
    /* x -= L21' xb: every thread owns a slice of the pivot columns */
    if (rs > 0) {
        i64 a, b; slice_of(0, ns, t, T, &a, &b);
        if (b > a) { const blasint brs = (blasint)rs, bc = (blasint)(b - a); h->dgemv("T", &brs, &bc, &mone, P + ns + a * ld, &bf, xb, &inc, &one, x + a, &inc); }
    }
#pragma omp barrier
    const i64 nkb = (ns + SOLVE_KB - 1) / SOLVE_KB;
    for (i64 kk = nkb - 1; kk >= 0; --kk) {
        const i64 k = kk * SOLVE_KB, kb = (ns - k < SOLVE_KB) ? ns - k : SOLVE_KB;
        const blasint bkb = (blasint)kb;
        if (t == 0) h->dtrsv("L", "T", "N", &bkb, P + k + k * ld, &bf, x + k, &inc);
#pragma omp barrier
        /* x[0, k) -= L[k .. k + kb, 0 .. k)' x[k .. k + kb): slices of the columns before the step */
        i64 a, b; slice_of(0, k, t, T, &a, &b);
        if (b > a) { const blasint bc = (blasint)(b - a); h->dgemv("T", &bkb, &bc, &mone, P + k + a * ld, &bf, x + k, &inc, &one, x + a, &inc); }
#pragma omp barrier
    }
}

/* K1SN_SOLVE_TEAM: 0 = every front on one thread; 1 (default) = the large fronts of a level with fewer than nthreads/2 fronts are split over the team
 * (a level of >= nthreads/2 fronts has one front per thread: measured on the 2 x 64-core host, 64 threads -- slicing each of the 64 3 500-column fronts
 * of config C4 into 64 row slices of ~70 rows made the solve 4x SLOWER than one front per thread: 568-byte runs per column, two barriers per
 * 256-column step across both sockets); 2 = split every large front (that experiment).  K1SN_TRACE=1: one line per level and direction on stderr. */
static int solve_team_mode(void) { const char *e = getenv("K1SN_SOLVE_TEAM"); return e ? atoi(e) : 1; }

static void solve_level(k1sn *h, i64 d, int backward) {
    const i64 a = h->level_ptr[d], b = h->level_ptr[d + 1];
    if (h->nthreads <= 1) {
        for (i64 t = a; t < b; ++t) { const i64 s = h->level_fronts[t]; if (backward) bwd_front(h, s); else fwd_front(h, s); }
        return;
    }
    const double t0 = now_s();
    const int mode = solve_team_mode();
    /* fronts of a level are sorted by decreasing work: the big ones form a prefix */
    i64 nbig = 0;
    if (mode == 2 || (mode == 1 && 2 * (b - a) < (i64)h->nthreads))
        while (a + nbig < b && h->f[h->level_fronts[a + nbig]] * h->ns[h->level_fronts[a + nbig]] >= SOLVE_BIG) ++nbig;
    const i64 nsmall = b - a - nbig;
    const long chunk = (nsmall > 64 * (i64)h->nthreads) ? 32 : 1;
#pragma omp parallel num_threads(h->nthreads)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        for (i64 q = 0; q < nbig; ++q) { const i64 s = h->level_fronts[a + q]; if (backward) bwd_front_team(h, s, t, T); else fwd_front_team(h, s, t, T); }
#pragma omp for schedule(dynamic, chunk)
        for (long q = (long)(a + nbig); q < (long)b; ++q) { const i64 s = h->level_fronts[q]; if (backward) bwd_front(h, s); else fwd_front(h, s); }
    }
    if (getenv("K1SN_TRACE"))
        fprintf(stderr, "k1sn %s level %lld: %lld fronts (%lld split over the team), largest %lld x %lld, %.4f s\n", backward ? "bwd" : "fwd", (long long)d,
                (long long)(b - a), (long long)nbig, (long long)h->f[h->level_fronts[a]], (long long)h->ns[h->level_fronts[a]], now_s() - t0);
}

int k1sn_solve(k1sn *h, double *dx, double *dy, const double *xi_p, const double *xi_d) {
    if (!h || !dx || !dy || !xi_p || !xi_d) return 2;
    if (!h->factored) return 7;
    const double t0 = now_s();
    const i64 m = h->m, n = h->n;
    double *xi = (double *)malloc((size_t)(m > 0 ? m : 1) * 8);
    if (!xi) return 3;
#pragma omp parallel for schedule(static, 1024) num_threads(h->nthreads)
    for (i64 i = 0; i < m; ++i) {                                                                /* spd.jl:56-57, row by row (columns ascending) */
        double v = xi_p[i];
        for (i64 q = h->Tp[i]; q < h->Tp[i + 1]; ++q) { const i64 j = h->Tj[q]; v += h->Tx[q] * (h->D[j] * xi_d[j]); }
        xi[i] = v;
    }
#pragma omp parallel for schedule(static, 4096) num_threads(h->nthreads)
    for (i64 ii = 0; ii < m; ++ii) h->xw[ii] = xi[h->perm[ii]];
    if (h->nthreads > 1) h->setnt(1);                                                            /* the team below is ours: BLAS calls stay single-threaded */
    for (i64 d = h->nlevels - 1; d >= 0; --d) solve_level(h, d, 0);                              /* spd.jl:61 */
    for (i64 d = 0; d < h->nlevels; ++d) solve_level(h, d, 1);
    if (h->nthreads > 1) h->setnt(h->nthreads);
#pragma omp parallel for schedule(static, 4096) num_threads(h->nthreads)
    for (i64 ii = 0; ii < m; ++ii) dy[h->perm[ii]] = h->xw[ii];
#pragma omp parallel for schedule(static, 1024) num_threads(h->nthreads)
    for (i64 j = 0; j < n; ++j) {                                                                /* spd.jl:64-66 */
        double sacc = 0.0;
        for (i64 p = h->Ap[j]; p < h->Ap[j + 1]; ++p) sacc += h->Ax[p] * dy[h->Ai[p]];
        dx[j] = h->D[j] * (sacc - xi_d[j]);
    }
    free(xi);
    h->t_solve = now_s() - t0;
    if (getenv("K1SN_TRACE")) fprintf(stderr, "k1sn solve: %.4f s on %d threads\n", h->t_solve, h->nthreads);
    return 0;
}

i64 k1sn_fail_col(const k1sn *h) { return h ? h->fail_col : -1; }
int k1sn_threads(const k1sn *h) { return h ? h->nthreads : 0; }
void k1sn_times(const k1sn *h, double *t3) { if (h && t3) { t3[0] = h->t_assemble; t3[1] = h->t_factor; t3[2] = h->t_solve; } }
/* the factor panels, same storage layout as tlpk_get_factor (front s: f x ns column-major, ld = f) */
int k1sn_get_factor(const k1sn *h, double *lval, i64 cap) {
    if (!h || !lval || cap < h->lval_len) return 2;
    memcpy(lval, h->Lval, (size_t)h->lval_len * 8);
    return 0;
}
const double *k1sn_factor_ptr(const k1sn *h) { return h ? h->Lval : NULL; }
/* diagonal of L in permuted order (m entries), for pivot reports: L_jj^2 is the pivot of column j.  After a failed
 * update the fronts factorised before the failure hold factor entries, the others their assembled values. */
int k1sn_get_diag(const k1sn *h, double *diag, i64 cap) {
    if (!h || !diag || cap < h->m) return 2;
    for (i64 s = 0; s < h->nf; ++s) {
        const double *P = h->Lval + h->loff[s];
        for (i64 j = 0; j < h->ns[s]; ++j) diag[h->col0[s] + j] = P[j + j * h->lda[s]];
    }
    return 0;
}
