/*
 * k1_oracle.c -- CPU restatement of Tulip.jl's normal-equations (K1) KKT path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the checker.
 * The product path (tulip.jl_amd/csrc, libtlpk.so) never links, loads or calls it.
 *
 * What it restates (citations are into /root/reference):
 *   src/KKT/Cholmod/spd.jl:5-20    setup   : pattern of S = A*A' + I is fixed once
 *   src/KKT/Cholmod/spd.jl:22-50   update! : D = 1/(theta+regP); S = A*D*A' + diag(regD);
 *                                            numeric Cholesky; not-SPD -> error
 *   src/KKT/Cholmod/spd.jl:52-70   solve!  : xi = xi_p + A*(D.*xi_d); dy = S \ xi;
 *                                            dx = D .* (A'*dy - xi_d)
 *   src/KKT/KKT.jl:65-100, src/KKT/systems.jl:34-54   definition of the K1 reduction
 *
 * The factorisation itself is third-party in the reference: SuiteSparse CHOLMOD through
 * Julia's SparseArrays stdlib (src/KKT/Cholmod/cholmod.jl:5; Project.toml pins only
 * `julia = "1.10"`, no Manifest => CHOLMOD version unpinned, SuiteSparse >= 7.2 implied).
 * CHOLMOD is absent from /root/reference and from this image, so its published algorithm is
 * restated here in its simplicial form: elimination tree (Liu 1990), symbolic column
 * structures by child merging, and a left-looking sparse column Cholesky P*S*P' = L*L'
 * (George & Liu 1981; Chen, Davis, Hager, Rajamanickam, ACM TOMS 35(3), 2008, section 2).
 * No pivot perturbation (CHOLMOD default dbound = 0), no iterative refinement (spd.jl:68).
 *
 * Pinning: tests/test_oracle.py checks this file against the reference's own fixture for the
 * path (test/KKT/Cholmod/cholmod.jl:3-16 + src/KKT/Test/test.jl:9-47: the 2x4 matrix with
 * all-ones data, residuals <= sqrt(eps)), against the JSON fixtures in tests/golden (dense augmented-system
 * solves of KKT.jl:70-75) and against dense numpy solves on random instances.  The reference
 * holds no golden vector of L, dx or dy, so bitwise parity with CHOLMOD is not defined;
 * parity = the same linear system solved to the reference's own tolerance.
 *
 * Plain C99, no dependencies.  Indices int64 on the boundary (Julia Int), base 0 or 1.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

typedef struct k1o {
    i64 m, n, nnzA;
    /* A in CSC (0-based) and its transpose (CSR of A) */
    i64 *Ap, *Ai; double *Ax;
    i64 *Tp, *Tj; double *Tx;      /* row i: columns Tj[Tp[i]..Tp[i+1]) */
    /* stored copies, spd.jl:36-38 */
    double *theta, *regP, *regD, *D;
    /* ordering: perm[new] = old, iperm[old] = new */
    i64 *perm, *iperm;
    /* lower triangle of C = P*S*P' in CSC, pattern fixed at setup */
    i64 *Cp, *Ci; double *Cx;
    /* factor L (CSC, diagonal first in each column, rows sorted) */
    i64 *parent, *Lp, *Li; double *Lx;
    i64 nnzL;
    /* work */
    double *w, *rhs;
    i64 *head, *next, *pos, *mark;
    int factored;
    i64 fail_col;                  /* column (in permuted order) of the failing pivot, or -1 */
} k1o;

#define K1O_OK 0
#define K1O_NOT_POSDEF 1
#define K1O_BADARG 2
#define K1O_NOMEM 3

static void *xcalloc(i64 cnt, size_t sz) { return calloc((size_t)(cnt > 0 ? cnt : 1), sz); }

static int cmp_i64(const void *a, const void *b) {
    i64 x = *(const i64 *)a, y = *(const i64 *)b;
    return (x > y) - (x < y);
}

void k1o_free(k1o *o) {
    if (!o) return;
    free(o->Ap); free(o->Ai); free(o->Ax); free(o->Tp); free(o->Tj); free(o->Tx);
    free(o->theta); free(o->regP); free(o->regD); free(o->D);
    free(o->perm); free(o->iperm); free(o->Cp); free(o->Ci); free(o->Cx);
    free(o->parent); free(o->Lp); free(o->Li); free(o->Lx);
    free(o->w); free(o->rhs); free(o->head); free(o->next); free(o->pos); free(o->mark);
    free(o);
}

/* Build the pattern of the lower triangle of C = P*(A*A' + I)*P'  (spd.jl:14: the pattern of S
 * is structural pattern(A*A') plus the diagonal and never changes afterwards). */
static int build_pattern(k1o *o) {
    const i64 m = o->m;
    i64 *mark = o->mark;
    for (i64 i = 0; i < m; ++i) mark[i] = -1;
    /* pass 1: count, pass 2: fill.  Column kk of C (permuted) <-> row k = perm[kk] of S. */
    for (int pass = 0; pass < 2; ++pass) {
        i64 nz = 0;
        for (i64 i = 0; i < m; ++i) mark[i] = -1;
        for (i64 kk = 0; kk < m; ++kk) {
            const i64 k = o->perm[kk];
            i64 start = nz;
            if (pass) o->Cp[kk] = nz;
            mark[kk] = kk;                       /* diagonal always present */
            if (pass) o->Ci[nz] = kk;
            ++nz;
            for (i64 p = o->Tp[k]; p < o->Tp[k + 1]; ++p) {
                const i64 j = o->Tj[p];
                for (i64 q = o->Ap[j]; q < o->Ap[j + 1]; ++q) {
                    const i64 ii = o->iperm[o->Ai[q]];
                    if (ii > kk && mark[ii] != kk) {
                        mark[ii] = kk;
                        if (pass) o->Ci[nz] = ii;
                        ++nz;
                    }
                }
            }
            if (pass) qsort(o->Ci + start, (size_t)(nz - start), sizeof(i64), cmp_i64);
        }
        if (!pass) {
            o->Cp = (i64 *)xcalloc(m + 1, sizeof(i64));
            o->Ci = (i64 *)xcalloc(nz, sizeof(i64));
            o->Cx = (double *)xcalloc(nz, sizeof(double));
            if (!o->Cp || !o->Ci || !o->Cx) return K1O_NOMEM;
        } else {
            o->Cp[m] = nz;
        }
    }
    return K1O_OK;
}

/* Elimination tree of C (lower CSC) -- Liu's algorithm with path compression, driven by
 * rows: needs, for each row i, the columns k < i with C[i,k] != 0, i.e. the transpose. */
static int symbolic(k1o *o) {
    const i64 m = o->m;
    const i64 nzC = o->Cp[m];
    i64 *Rp = (i64 *)xcalloc(m + 1, sizeof(i64));
    i64 *Rj = (i64 *)xcalloc(nzC, sizeof(i64));
    i64 *anc = (i64 *)xcalloc(m, sizeof(i64));
    i64 *cnt = (i64 *)xcalloc(m + 1, sizeof(i64));
    if (!Rp || !Rj || !anc || !cnt) return K1O_NOMEM;
    for (i64 k = 0; k < m; ++k)
        for (i64 p = o->Cp[k]; p < o->Cp[k + 1]; ++p) Rp[o->Ci[p] + 1]++;
    for (i64 i = 0; i < m; ++i) Rp[i + 1] += Rp[i];
    for (i64 k = 0; k < m; ++k)
        for (i64 p = o->Cp[k]; p < o->Cp[k + 1]; ++p) {
            i64 i = o->Ci[p];
            Rj[Rp[i] + cnt[i]++] = k;            /* columns in increasing order */
        }
    o->parent = (i64 *)xcalloc(m, sizeof(i64));
    for (i64 i = 0; i < m; ++i) {
        o->parent[i] = -1; anc[i] = -1;
        for (i64 p = Rp[i]; p < Rp[i + 1]; ++p) {
            i64 k = Rj[p];
            while (k != -1 && k < i) {
                i64 nxt = anc[k];
                anc[k] = i;
                if (nxt == -1) o->parent[k] = i;
                k = nxt;
            }
        }
    }
    /* column structures: struct(j) = rows(C[:,j]) U (U_{c child of j} struct(c) \ {c}) */
    i64 **cols = (i64 **)xcalloc(m, sizeof(i64 *));
    i64 *len = (i64 *)xcalloc(m, sizeof(i64));
    i64 *chead = (i64 *)xcalloc(m, sizeof(i64)), *cnext = (i64 *)xcalloc(m, sizeof(i64));
    for (i64 i = 0; i < m; ++i) { chead[i] = -1; o->mark[i] = -1; }
    for (i64 j = m - 1; j >= 0; --j)
        if (o->parent[j] >= 0) { cnext[j] = chead[o->parent[j]]; chead[o->parent[j]] = j; }
    i64 *tmp = (i64 *)xcalloc(m, sizeof(i64));
    i64 nnzL = 0;
    for (i64 j = 0; j < m; ++j) {
        i64 c = 0;
        for (i64 p = o->Cp[j]; p < o->Cp[j + 1]; ++p) {
            i64 i = o->Ci[p];
            if (o->mark[i] != j) { o->mark[i] = j; tmp[c++] = i; }
        }
        for (i64 ch = chead[j]; ch != -1; ch = cnext[ch]) {
            for (i64 t = 1; t < len[ch]; ++t) {  /* skip the diagonal entry of the child */
                i64 i = cols[ch][t];
                if (o->mark[i] != j) { o->mark[i] = j; tmp[c++] = i; }
            }
        }
        qsort(tmp, (size_t)c, sizeof(i64), cmp_i64);
        cols[j] = (i64 *)malloc((size_t)c * sizeof(i64));
        if (!cols[j]) return K1O_NOMEM;
        memcpy(cols[j], tmp, (size_t)c * sizeof(i64));
        len[j] = c;
        nnzL += c;
    }
    o->nnzL = nnzL;
    o->Lp = (i64 *)xcalloc(m + 1, sizeof(i64));
    o->Li = (i64 *)xcalloc(nnzL, sizeof(i64));
    o->Lx = (double *)xcalloc(nnzL, sizeof(double));
    if (!o->Lp || !o->Li || !o->Lx) return K1O_NOMEM;
    for (i64 j = 0; j < m; ++j) {
        o->Lp[j + 1] = o->Lp[j] + len[j];
        memcpy(o->Li + o->Lp[j], cols[j], (size_t)len[j] * sizeof(i64));
        free(cols[j]);
    }
    free(cols); free(len); free(chead); free(cnext); free(tmp);
    free(Rp); free(Rj); free(anc); free(cnt);
    return K1O_OK;
}

/* setup -- spd.jl:5-20.  perm may be NULL (natural order); perm[new] = old, in index_base. */
int k1o_setup(k1o **out, i64 m, i64 n, const i64 *colptr, const i64 *rowval, const double *nzval,
              int index_base, const i64 *perm) {
    if (!out || m < 0 || n < 0 || !colptr || (index_base != 0 && index_base != 1)) return K1O_BADARG;
    k1o *o = (k1o *)calloc(1, sizeof(k1o));
    if (!o) return K1O_NOMEM;
    o->m = m; o->n = n; o->fail_col = -1;
    const i64 nnz = colptr[n] - index_base;
    o->nnzA = nnz;
    o->Ap = (i64 *)xcalloc(n + 1, sizeof(i64));
    o->Ai = (i64 *)xcalloc(nnz, sizeof(i64));
    o->Ax = (double *)xcalloc(nnz, sizeof(double));
    o->Tp = (i64 *)xcalloc(m + 1, sizeof(i64));
    o->Tj = (i64 *)xcalloc(nnz, sizeof(i64));
    o->Tx = (double *)xcalloc(nnz, sizeof(double));
    o->theta = (double *)xcalloc(n, sizeof(double));
    o->regP = (double *)xcalloc(n, sizeof(double));
    o->regD = (double *)xcalloc(m, sizeof(double));
    o->D = (double *)xcalloc(n, sizeof(double));
    o->perm = (i64 *)xcalloc(m, sizeof(i64));
    o->iperm = (i64 *)xcalloc(m, sizeof(i64));
    o->w = (double *)xcalloc(m, sizeof(double));
    o->rhs = (double *)xcalloc(m, sizeof(double));
    o->head = (i64 *)xcalloc(m, sizeof(i64));
    o->next = (i64 *)xcalloc(m, sizeof(i64));
    o->pos = (i64 *)xcalloc(m, sizeof(i64));
    o->mark = (i64 *)xcalloc(m, sizeof(i64));
    for (i64 j = 0; j <= n; ++j) o->Ap[j] = colptr[j] - index_base;
    for (i64 p = 0; p < nnz; ++p) {
        o->Ai[p] = rowval[p] - index_base;
        o->Ax[p] = nzval[p];
        if (o->Ai[p] < 0 || o->Ai[p] >= m) { k1o_free(o); return K1O_BADARG; }
        o->Tp[o->Ai[p] + 1]++;
    }
    for (i64 i = 0; i < m; ++i) o->Tp[i + 1] += o->Tp[i];
    {
        i64 *c = (i64 *)xcalloc(m, sizeof(i64));
        for (i64 j = 0; j < n; ++j)
            for (i64 p = o->Ap[j]; p < o->Ap[j + 1]; ++p) {
                i64 i = o->Ai[p];
                i64 q = o->Tp[i] + c[i]++;
                o->Tj[q] = j; o->Tx[q] = o->Ax[p];
            }
        free(c);
    }
    for (i64 i = 0; i < m; ++i) o->iperm[i] = -1;
    for (i64 i = 0; i < m; ++i) {
        i64 old = perm ? perm[i] - index_base : i;
        if (old < 0 || old >= m || o->iperm[old] != -1) { k1o_free(o); return K1O_BADARG; }
        o->perm[i] = old; o->iperm[old] = i;
    }
    /* spd.jl:8-10: theta = regP = regD = 1 */
    for (i64 j = 0; j < n; ++j) { o->theta[j] = 1.0; o->regP[j] = 1.0; }
    for (i64 i = 0; i < m; ++i) o->regD[i] = 1.0;
    int rc = build_pattern(o);
    if (rc == K1O_OK) rc = symbolic(o);
    if (rc != K1O_OK) { k1o_free(o); return rc; }
    *out = o;
    return K1O_OK;
}

/* Numeric values of C = P*(A*D*A' + diag(regD))*P' on the fixed pattern -- spd.jl:42-43. */
static void form_normal_equations(k1o *o) {
    const i64 m = o->m;
    double *w = o->w;
    for (i64 i = 0; i < m; ++i) w[i] = 0.0;
    for (i64 kk = 0; kk < m; ++kk) {
        const i64 k = o->perm[kk];
        for (i64 p = o->Tp[k]; p < o->Tp[k + 1]; ++p) {
            const i64 j = o->Tj[p];
            const double akj_d = o->Tx[p] * o->D[j];
            for (i64 q = o->Ap[j]; q < o->Ap[j + 1]; ++q) {
                const i64 ii = o->iperm[o->Ai[q]];
                if (ii >= kk) w[ii] += akj_d * o->Ax[q];
            }
        }
        w[kk] += o->regD[k];
        for (i64 p = o->Cp[kk]; p < o->Cp[kk + 1]; ++p) { o->Cx[p] = w[o->Ci[p]]; w[o->Ci[p]] = 0.0; }
    }
}

/* Left-looking sparse column Cholesky on the fixed structure.  Returns K1O_NOT_POSDEF at the
 * first pivot that is <= 0 or NaN (spd.jl:46-47: cholesky!(check=false) + issuccess). */
static int numeric_cholesky(k1o *o) {
    const i64 m = o->m;
    double *w = o->w;
    i64 *head = o->head, *next = o->next, *pos = o->pos;
    for (i64 i = 0; i < m; ++i) { head[i] = -1; w[i] = 0.0; }
    o->fail_col = -1;
    for (i64 j = 0; j < m; ++j) {
        for (i64 p = o->Cp[j]; p < o->Cp[j + 1]; ++p) w[o->Ci[p]] = o->Cx[p];
        i64 k = head[j];
        while (k != -1) {
            const i64 knext = next[k];
            const i64 p0 = pos[k];               /* L[j,k] sits at p0 */
            const double ljk = o->Lx[p0];
            for (i64 p = p0; p < o->Lp[k + 1]; ++p) w[o->Li[p]] -= o->Lx[p] * ljk;
            if (p0 + 1 < o->Lp[k + 1]) {
                pos[k] = p0 + 1;
                const i64 r = o->Li[p0 + 1];
                next[k] = head[r]; head[r] = k;
            }
            k = knext;
        }
        const double d = w[j];
        if (!(d > 0.0)) {                        /* also catches NaN */
            o->fail_col = j;
            for (i64 p = o->Lp[j]; p < o->Lp[j + 1]; ++p) w[o->Li[p]] = 0.0;
            return K1O_NOT_POSDEF;
        }
        const double ljj = sqrt(d);
        const i64 s = o->Lp[j];
        o->Lx[s] = ljj; w[j] = 0.0;
        for (i64 p = s + 1; p < o->Lp[j + 1]; ++p) { o->Lx[p] = w[o->Li[p]] / ljj; w[o->Li[p]] = 0.0; }
        if (s + 1 < o->Lp[j + 1]) {
            pos[j] = s + 1;
            const i64 r = o->Li[s + 1];
            next[j] = head[r]; head[r] = j;
        }
    }
    return K1O_OK;
}

/* update! -- spd.jl:22-50 */
int k1o_update(k1o *o, const double *theta, const double *regP, const double *regD) {
    if (!o || !theta || !regP || !regD) return K1O_BADARG;
    memcpy(o->theta, theta, (size_t)o->n * sizeof(double));      /* spd.jl:36-38 */
    memcpy(o->regP, regP, (size_t)o->n * sizeof(double));
    memcpy(o->regD, regD, (size_t)o->m * sizeof(double));
    for (i64 j = 0; j < o->n; ++j) o->D[j] = 1.0 / (o->theta[j] + o->regP[j]);   /* spd.jl:42 */
    form_normal_equations(o);                                     /* spd.jl:43 */
    o->factored = 0;
    int rc = numeric_cholesky(o);                                 /* spd.jl:46-47 */
    if (rc == K1O_OK) o->factored = 1;
    return rc;
}

/* solve! -- spd.jl:52-70.  dx, dy fully overwritten; xi_p, xi_d read-only. */
int k1o_solve(k1o *o, double *dx, double *dy, const double *xi_p, const double *xi_d) {
    if (!o || !dx || !dy || !xi_p || !xi_d) return K1O_BADARG;
    if (!o->factored) return K1O_NOT_POSDEF;
    const i64 m = o->m, n = o->n;
    double *x = o->rhs, *xi = o->w;
    /* xi = xi_p + A*(D.*xi_d)   spd.jl:56-57 */
    for (i64 i = 0; i < m; ++i) {
        double s = xi_p[i];
        for (i64 p = o->Tp[i]; p < o->Tp[i + 1]; ++p) s += o->Tx[p] * (o->D[o->Tj[p]] * xi_d[o->Tj[p]]);
        xi[i] = s;
    }
    for (i64 i = 0; i < m; ++i) x[i] = xi[o->perm[i]];
    for (i64 i = 0; i < m; ++i) xi[i] = 0.0;     /* restore the all-zero invariant of w */
    /* dy = S \ xi    spd.jl:61 : L z = P xi ; L' y = z ; dy = P' y */
    for (i64 j = 0; j < m; ++j) {
        const i64 s = o->Lp[j];
        x[j] /= o->Lx[s];
        const double xj = x[j];
        for (i64 p = s + 1; p < o->Lp[j + 1]; ++p) x[o->Li[p]] -= o->Lx[p] * xj;
    }
    for (i64 j = m - 1; j >= 0; --j) {
        const i64 s = o->Lp[j];
        double t = x[j];
        for (i64 p = s + 1; p < o->Lp[j + 1]; ++p) t -= o->Lx[p] * x[o->Li[p]];
        x[j] = t / o->Lx[s];
    }
    for (i64 i = 0; i < m; ++i) dy[o->perm[i]] = x[i];
    /* dx = D .* (A'*dy - xi_d)   spd.jl:64-66 */
    for (i64 j = 0; j < n; ++j) {
        double s = 0.0;
        for (i64 p = o->Ap[j]; p < o->Ap[j + 1]; ++p) s += o->Ax[p] * dy[o->Ai[p]];
        dx[j] = o->D[j] * (s - xi_d[j]);
    }
    return K1O_OK;
}

/* ---- inspection helpers for the tests ---- */
i64 k1o_nnzS(const k1o *o) { return o->Cp[o->m]; }
i64 k1o_nnzL(const k1o *o) { return o->nnzL; }
i64 k1o_fail_col(const k1o *o) { return o->fail_col; }
double k1o_flops(const k1o *o) {                 /* sum_j l_j^2, CHOLMOD's `fl` convention */
    double f = 0.0;
    for (i64 j = 0; j < o->m; ++j) { double l = (double)(o->Lp[j + 1] - o->Lp[j]); f += l * l; }
    return f;
}
/* copy out the lower triangle of the permuted S (CSC, 0-based) */
void k1o_get_S(const k1o *o, i64 *colptr, i64 *rowval, double *val) {
    memcpy(colptr, o->Cp, (size_t)(o->m + 1) * sizeof(i64));
    memcpy(rowval, o->Ci, (size_t)o->Cp[o->m] * sizeof(i64));
    memcpy(val, o->Cx, (size_t)o->Cp[o->m] * sizeof(double));
}
void k1o_get_L(const k1o *o, i64 *colptr, i64 *rowval, double *val) {
    memcpy(colptr, o->Lp, (size_t)(o->m + 1) * sizeof(i64));
    memcpy(rowval, o->Li, (size_t)o->nnzL * sizeof(i64));
    memcpy(val, o->Lx, (size_t)o->nnzL * sizeof(double));
}
