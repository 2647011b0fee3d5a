/*
 * k2_oracle.c -- CPU restatement of Tulip.jl's augmented-system (K2) KKT path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as k1_oracle.c).  It is the oracle of the
 * NEXT row of the hot-path contract (a device LDL' of the quasi-definite
 * augmented matrix); nothing in tulip.jl_amd/ implements K2 yet, and nothing in the product may
 * link, load or call this file.
 *
 * What it restates (citations are into /root/reference):
 *   src/KKT/Cholmod/sqd.jl:5-22    setup   : K = [ -Theta^-1  A' ; A  I ] (pattern fixed once)
 *   src/KKT/Cholmod/sqd.jl:24-54   update! : diag(K) <- ( -(theta + regP), regD ); numeric LDL'
 *   src/KKT/Cholmod/sqd.jl:56-74   solve!  : [dx; dy] = K \ [xi_d; xi_p]
 *   src/KKT/KKT.jl:65-100, src/KKT/systems.jl:12-31   definition of the augmented system
 *
 * The factorisation is third-party in the reference (SuiteSparse CHOLMOD `ldlt` through Julia's
 * SparseArrays stdlib, src/KKT/Cholmod/cholmod.jl:5; version unpinned: Project.toml has only
 * `julia = "1.10"`).  Its published algorithm is restated in simplicial form: elimination tree
 * (Liu 1990), column structures by child merging, left-looking column LDL' with a unit lower
 * triangular L and a diagonal D without pivoting -- valid for symmetric quasi-definite matrices
 * under any symmetric permutation (Vanderbei 1995).  A zero (or NaN) pivot is an error
 * (ZeroPivotException in the reference, caught by the IPM like PosDefException, HSD/step.jl:40).
 * The sign pattern of D is checked as well: the first n original indices must give negative pivots,
 * the last m positive ones.
 *
 * Pinning: tests/test_oracle.py (K2 section) -- the reference's own fixture for this path
 * (test/KKT/Cholmod/cholmod.jl:3-16 runs KKT.run_ls_tests on CholmodSolver{Float64,K2}), the JSON
 * fixtures of tests/golden (dense augmented-system solves of KKT.jl:70-75: the K1 and K2 paths
 * solve the SAME system, so the same golden dx, dy apply), and the K1 oracle on random instances.
 *
 * Plain C99, no dependencies.  Indices int64 on the boundary (Julia Int), base 0 or 1.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

typedef struct k2o {
    i64 m, n, N, nnzA;
    i64 *Ap, *Ai; double *Ax;          /* A in CSC, 0-based */
    double *theta, *regP, *regD;       /* stored copies, sqd.jl:40-42 */
    i64 *perm, *iperm;                 /* perm[new] = old over the N = n + m nodes (variables first) */
    i64 *Cp, *Ci; double *Cx;          /* lower triangle of P*K*P' in CSC, diagonal first, rows sorted */
    i64 *dpos;                         /* position in Cx of the diagonal entry of original node v */
    i64 *parent, *Lp, *Li; double *Lx; /* unit lower L (diagonal entry stored first, value 1) */
    double *d;                         /* D */
    i64 nnzL;
    double *w, *x;
    i64 *head, *next, *pos, *mark;
    int factored;
    i64 fail_col;
} k2o;

#define K2O_OK 0
#define K2O_ZERO_PIVOT 1
#define K2O_BADARG 2
#define K2O_NOMEM 3

static void *xcalloc(i64 cnt, size_t sz) { return calloc((size_t)(cnt > 0 ? cnt : 1), sz); }
static int cmp_i64(const void *a, const void *b) {
    const i64 x = *(const i64 *)a, y = *(const i64 *)b;
    return (x > y) - (x < y);
}

void k2o_free(k2o *o) {
    if (!o) return;
    free(o->Ap); free(o->Ai); free(o->Ax); free(o->theta); free(o->regP); free(o->regD);
    free(o->perm); free(o->iperm); free(o->Cp); free(o->Ci); free(o->Cx); free(o->dpos);
    free(o->parent); free(o->Lp); free(o->Li); free(o->Lx); free(o->d);
    free(o->w); free(o->x); free(o->head); free(o->next); free(o->pos); free(o->mark);
    free(o);
}

/* Lower pattern of C = P*K*P': node v < n is variable v (column of A), node n + i is constraint i.
 * K[n+i, j] = A[i, j]; the diagonal is always stored (sqd.jl:13-16). */
static int build_pattern(k2o *o) {
    const i64 N = o->N, n = o->n;
    i64 *cnt = (i64 *)xcalloc(N + 1, sizeof(i64));
    if (!cnt) return K2O_NOMEM;
    for (i64 v = 0; v < N; ++v) cnt[o->iperm[v] + 1]++;                /* diagonal */
    for (i64 j = 0; j < n; ++j)
        for (i64 p = o->Ap[j]; p < o->Ap[j + 1]; ++p) {
            const i64 a = o->iperm[j], b = o->iperm[n + o->Ai[p]];
            cnt[(a < b ? a : b) + 1]++;
        }
    for (i64 k = 0; k < N; ++k) cnt[k + 1] += cnt[k];
    const i64 nz = cnt[N];
    o->Cp = (i64 *)xcalloc(N + 1, sizeof(i64));
    o->Ci = (i64 *)xcalloc(nz, sizeof(i64));
    o->Cx = (double *)xcalloc(nz, sizeof(double));
    o->dpos = (i64 *)xcalloc(N, sizeof(i64));
    i64 *cur = (i64 *)xcalloc(N, sizeof(i64));
    if (!o->Cp || !o->Ci || !o->Cx || !o->dpos || !cur) { free(cnt); free(cur); return K2O_NOMEM; }
    memcpy(o->Cp, cnt, (size_t)(N + 1) * sizeof(i64));
    for (i64 k = 0; k < N; ++k) cur[k] = o->Cp[k];
    for (i64 v = 0; v < N; ++v) { const i64 k = o->iperm[v]; o->Ci[cur[k]++] = k; }   /* diagonal first */
    for (i64 j = 0; j < n; ++j)
        for (i64 p = o->Ap[j]; p < o->Ap[j + 1]; ++p) {
            const i64 a = o->iperm[j], b = o->iperm[n + o->Ai[p]];
            const i64 col = a < b ? a : b, row = a < b ? b : a;
            o->Ci[cur[col]++] = row;
        }
    for (i64 k = 0; k < N; ++k) qsort(o->Ci + o->Cp[k] + 1, (size_t)(o->Cp[k + 1] - o->Cp[k] - 1), sizeof(i64), cmp_i64);
    for (i64 v = 0; v < N; ++v) o->dpos[v] = o->Cp[o->iperm[v]];
    free(cnt); free(cur);
    return K2O_OK;
}

/* values of the off-diagonal part (A) -- constant over the life of the handle */
static void fill_offdiag(k2o *o) {
    const i64 n = o->n;
    for (i64 j = 0; j < n; ++j)
        for (i64 p = o->Ap[j]; p < o->Ap[j + 1]; ++p) {
            const i64 a = o->iperm[j], b = o->iperm[n + o->Ai[p]];
            const i64 col = a < b ? a : b, row = a < b ? b : a;
            /* duplicates (i, j) in A are summed */
            i64 lo = o->Cp[col] + 1, hi = o->Cp[col + 1];
            while (lo < hi) { const i64 mid = (lo + hi) >> 1; if (o->Ci[mid] < row) lo = mid + 1; else hi = mid; }
            o->Cx[lo] += o->Ax[p];
        }
}

/* elimination tree + column structures, as in k1_oracle.c */
static int symbolic(k2o *o) {
    const i64 N = o->N, nzC = o->Cp[N];
    i64 *Rp = (i64 *)xcalloc(N + 1, sizeof(i64)), *Rj = (i64 *)xcalloc(nzC, sizeof(i64));
    i64 *anc = (i64 *)xcalloc(N, sizeof(i64)), *cnt = (i64 *)xcalloc(N + 1, sizeof(i64));
    if (!Rp || !Rj || !anc || !cnt) return K2O_NOMEM;
    for (i64 k = 0; k < N; ++k)
        for (i64 p = o->Cp[k]; p < o->Cp[k + 1]; ++p) Rp[o->Ci[p] + 1]++;
    for (i64 i = 0; i < N; ++i) Rp[i + 1] += Rp[i];
    for (i64 k = 0; k < N; ++k)
        for (i64 p = o->Cp[k]; p < o->Cp[k + 1]; ++p) { const i64 i = o->Ci[p]; Rj[Rp[i] + cnt[i]++] = k; }
    o->parent = (i64 *)xcalloc(N, sizeof(i64));
    for (i64 i = 0; i < N; ++i) {
        o->parent[i] = -1; anc[i] = -1;
        for (i64 p = Rp[i]; p < Rp[i + 1]; ++p) {
            i64 k = Rj[p];
            while (k != -1 && k < i) {
                const i64 nxt = anc[k];
                anc[k] = i;
                if (nxt == -1) o->parent[k] = i;
                k = nxt;
            }
        }
    }
    i64 **cols = (i64 **)xcalloc(N, sizeof(i64 *));
    i64 *len = (i64 *)xcalloc(N, sizeof(i64)), *chead = (i64 *)xcalloc(N, sizeof(i64)), *cnext = (i64 *)xcalloc(N, sizeof(i64));
    i64 *tmp = (i64 *)xcalloc(N, sizeof(i64));
    for (i64 i = 0; i < N; ++i) { chead[i] = -1; o->mark[i] = -1; }
    for (i64 j = N - 1; j >= 0; --j)
        if (o->parent[j] >= 0) { cnext[j] = chead[o->parent[j]]; chead[o->parent[j]] = j; }
    i64 nnzL = 0;
    for (i64 j = 0; j < N; ++j) {
        i64 c = 0;
        for (i64 p = o->Cp[j]; p < o->Cp[j + 1]; ++p) {
            const i64 i = o->Ci[p];
            if (o->mark[i] != j) { o->mark[i] = j; tmp[c++] = i; }
        }
        for (i64 ch = chead[j]; ch != -1; ch = cnext[ch])
            for (i64 t = 1; t < len[ch]; ++t) {
                const i64 i = cols[ch][t];
                if (o->mark[i] != j) { o->mark[i] = j; tmp[c++] = i; }
            }
        qsort(tmp, (size_t)c, sizeof(i64), cmp_i64);
        cols[j] = (i64 *)malloc((size_t)(c > 0 ? c : 1) * sizeof(i64));
        if (!cols[j]) return K2O_NOMEM;
        memcpy(cols[j], tmp, (size_t)c * sizeof(i64));
        len[j] = c; nnzL += c;
    }
    o->nnzL = nnzL;
    o->Lp = (i64 *)xcalloc(N + 1, sizeof(i64));
    o->Li = (i64 *)xcalloc(nnzL, sizeof(i64));
    o->Lx = (double *)xcalloc(nnzL, sizeof(double));
    if (!o->Lp || !o->Li || !o->Lx) return K2O_NOMEM;
    for (i64 j = 0; j < N; ++j) {
        o->Lp[j + 1] = o->Lp[j] + len[j];
        memcpy(o->Li + o->Lp[j], cols[j], (size_t)len[j] * sizeof(i64));
        free(cols[j]);
    }
    free(cols); free(len); free(chead); free(cnext); free(tmp); free(Rp); free(Rj); free(anc); free(cnt);
    return K2O_OK;
}

/* setup -- sqd.jl:5-22.  perm may be NULL: variables in their order, then constraints (eliminating
 * all variables first leaves the normal equations as the Schur complement);
 * perm[new] = old over 0..n+m-1 (old < n: variable, old >= n: constraint old - n), in index_base. */
int k2o_setup(k2o **out, i64 m, i64 n, const i64 *colptr, const i64 *rowval, const double *nzval,
              int index_base, const i64 *perm) {
    if (!out || m < 0 || n < 0 || !colptr || (index_base != 0 && index_base != 1)) return K2O_BADARG;
    k2o *o = (k2o *)calloc(1, sizeof(k2o));
    if (!o) return K2O_NOMEM;
    const i64 N = n + m, nnz = colptr[n] - index_base;
    o->m = m; o->n = n; o->N = N; o->nnzA = nnz; o->fail_col = -1;
    o->Ap = (i64 *)xcalloc(n + 1, sizeof(i64)); o->Ai = (i64 *)xcalloc(nnz, sizeof(i64)); o->Ax = (double *)xcalloc(nnz, sizeof(double));
    o->theta = (double *)xcalloc(n, sizeof(double)); o->regP = (double *)xcalloc(n, sizeof(double)); o->regD = (double *)xcalloc(m, sizeof(double));
    o->perm = (i64 *)xcalloc(N, sizeof(i64)); o->iperm = (i64 *)xcalloc(N, sizeof(i64));
    o->d = (double *)xcalloc(N, sizeof(double)); o->w = (double *)xcalloc(N, sizeof(double)); o->x = (double *)xcalloc(N, sizeof(double));
    o->head = (i64 *)xcalloc(N, sizeof(i64)); o->next = (i64 *)xcalloc(N, sizeof(i64));
    o->pos = (i64 *)xcalloc(N, sizeof(i64)); o->mark = (i64 *)xcalloc(N, sizeof(i64));
    for (i64 j = 0; j <= n; ++j) o->Ap[j] = colptr[j] - index_base;
    for (i64 p = 0; p < nnz; ++p) {
        o->Ai[p] = rowval[p] - index_base; o->Ax[p] = nzval[p];
        if (o->Ai[p] < 0 || o->Ai[p] >= m) { k2o_free(o); return K2O_BADARG; }
    }
    for (i64 v = 0; v < N; ++v) o->iperm[v] = -1;
    for (i64 k = 0; k < N; ++k) {
        const i64 old = perm ? perm[k] - index_base : k;
        if (old < 0 || old >= N || o->iperm[old] != -1) { k2o_free(o); return K2O_BADARG; }
        o->perm[k] = old; o->iperm[old] = k;
    }
    for (i64 j = 0; j < n; ++j) { o->theta[j] = 1.0; o->regP[j] = 1.0; }     /* sqd.jl:8-10 */
    for (i64 i = 0; i < m; ++i) o->regD[i] = 1.0;
    int rc = build_pattern(o);
    if (rc == K2O_OK) { fill_offdiag(o); rc = symbolic(o); }
    if (rc != K2O_OK) { k2o_free(o); return rc; }
    *out = o;
    return K2O_OK;
}

/* left-looking column LDL', unit lower L */
static int numeric_ldlt(k2o *o) {
    const i64 N = o->N;
    double *w = o->w;
    i64 *head = o->head, *next = o->next, *pos = o->pos;
    for (i64 i = 0; i < N; ++i) { head[i] = -1; w[i] = 0.0; }
    o->fail_col = -1;
    for (i64 j = 0; j < N; ++j) {
        for (i64 p = o->Cp[j]; p < o->Cp[j + 1]; ++p) w[o->Ci[p]] = o->Cx[p];
        i64 k = head[j];
        while (k != -1) {
            const i64 knext = next[k], p0 = pos[k];
            const double ljk_dk = o->Lx[p0] * o->d[k];            /* L[j,k] * d_k */
            for (i64 p = p0; p < o->Lp[k + 1]; ++p) w[o->Li[p]] -= o->Lx[p] * ljk_dk;
            if (p0 + 1 < o->Lp[k + 1]) {
                pos[k] = p0 + 1;
                const i64 r = o->Li[p0 + 1];
                next[k] = head[r]; head[r] = k;
            }
            k = knext;
        }
        const double dj = w[j];
        const int want_negative = o->perm[j] < o->n;               /* variable node: -(theta + regP) block */
        if (!(dj != 0.0) || (want_negative ? !(dj < 0.0) : !(dj > 0.0))) {   /* zero, NaN, or not quasi-definite */
            o->fail_col = j;
            for (i64 p = o->Lp[j]; p < o->Lp[j + 1]; ++p) w[o->Li[p]] = 0.0;
            return K2O_ZERO_PIVOT;
        }
        const i64 s = o->Lp[j];
        o->d[j] = dj; o->Lx[s] = 1.0; w[j] = 0.0;
        for (i64 p = s + 1; p < o->Lp[j + 1]; ++p) { o->Lx[p] = w[o->Li[p]] / dj; w[o->Li[p]] = 0.0; }
        if (s + 1 < o->Lp[j + 1]) {
            pos[j] = s + 1;
            const i64 r = o->Li[s + 1];
            next[j] = head[r]; head[r] = j;
        }
    }
    return K2O_OK;
}

/* update! -- sqd.jl:24-54 */
int k2o_update(k2o *o, const double *theta, const double *regP, const double *regD) {
    if (!o || !theta || !regP || !regD) return K2O_BADARG;
    memcpy(o->theta, theta, (size_t)o->n * sizeof(double));          /* sqd.jl:40-42 */
    memcpy(o->regP, regP, (size_t)o->n * sizeof(double));
    memcpy(o->regD, regD, (size_t)o->m * sizeof(double));
    for (i64 j = 0; j < o->n; ++j) o->Cx[o->dpos[j]] = -o->theta[j] - o->regP[j];     /* sqd.jl:46-49 */
    for (i64 i = 0; i < o->m; ++i) o->Cx[o->dpos[o->n + i]] = o->regD[i];            /* sqd.jl:50-53 */
    o->factored = 0;
    const int rc = numeric_ldlt(o);                                   /* sqd.jl:55 */
    if (rc == K2O_OK) o->factored = 1;
    return rc;
}

/* solve! -- sqd.jl:56-74.  dx, dy fully overwritten; xi_p, xi_d read-only. */
int k2o_solve(k2o *o, double *dx, double *dy, const double *xi_p, const double *xi_d) {
    if (!o || !dx || !dy || !xi_p || !xi_d) return K2O_BADARG;
    if (!o->factored) return K2O_ZERO_PIVOT;
    const i64 N = o->N, n = o->n;
    double *x = o->x;
    for (i64 k = 0; k < N; ++k) { const i64 v = o->perm[k]; x[k] = v < n ? xi_d[v] : xi_p[v - n]; }   /* sqd.jl:60-61 */
    for (i64 j = 0; j < N; ++j) {                                     /* L z = b */
        const double xj = x[j];
        for (i64 p = o->Lp[j] + 1; p < o->Lp[j + 1]; ++p) x[o->Li[p]] -= o->Lx[p] * xj;
    }
    for (i64 j = 0; j < N; ++j) x[j] /= o->d[j];                      /* D y = z */
    for (i64 j = N - 1; j >= 0; --j) {                                /* L' x = y */
        double t = x[j];
        for (i64 p = o->Lp[j] + 1; p < o->Lp[j + 1]; ++p) t -= o->Lx[p] * x[o->Li[p]];
        x[j] = t;
    }
    for (i64 k = 0; k < N; ++k) { const i64 v = o->perm[k]; if (v < n) dx[v] = x[k]; else dy[v - n] = x[k]; }   /* sqd.jl:68-69 */
    return K2O_OK;
}

i64 k2o_nnzK(const k2o *o) { return o->Cp[o->N]; }
i64 k2o_nnzL(const k2o *o) { return o->nnzL; }
i64 k2o_fail_col(const k2o *o) { return o->fail_col; }
void k2o_get_D(const k2o *o, double *d) { memcpy(d, o->d, (size_t)o->N * sizeof(double)); }
