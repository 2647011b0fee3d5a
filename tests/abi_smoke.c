/* abi_smoke.c -- drives libtlpk.so exactly as the Julia glue's `ccall`s do
 * (tulip.jl_amd/julia/libtlpk.jl): a plain C caller, 1-based Int64 CSC arrays
 * (Julia's SparseMatrixCSC{Float64,Int}), host vectors, return codes only.  Compiled and run by
 * tests/test_abi.py (gcc, no HIP headers, no torch): `abi_smoke <path/libtlpk.so> cpu|gpu`.
 *   cpu: analyse-only handle (device = -1): tlpk_create / tlpk_info / tlpk_get_perm work, every
 *        numeric call returns TLPK_NO_DEVICE (no CPU fallback), tlpk_destroy.
 *   gpu: KKT.setup -> update! -> solve! -> the residual identities of
 *        /root/reference/src/KKT/Test/test.jl:39-43 on the reference's own 2 x 4 fixture
 *        (test/KKT/Cholmod/cholmod.jl:3-6) and on a 1-based random LP matrix; PosDef retry contract.
 * Test infrastructure. */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/tlpk.h"

#define LOAD(name) do { *(void **)(&p_##name) = dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; } } while (0)
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "abi_smoke: check failed at line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

static void (*p_tlpk_default_options)(tlpk_options *);
static int (*p_tlpk_create)(tlpk_handle **, int64_t, int64_t, const int64_t *, const int64_t *, const double *, int, const tlpk_options *);
static void (*p_tlpk_destroy)(tlpk_handle *);
static int (*p_tlpk_update)(tlpk_handle *, const double *, const double *, const double *);
static int (*p_tlpk_solve)(tlpk_handle *, double *, double *, const double *, const double *);
static int (*p_tlpk_info)(const tlpk_handle *, tlpk_stats *);
static int (*p_tlpk_get_perm)(const tlpk_handle *, int64_t *);
static const char *(*p_tlpk_strerror)(int);
static const char *(*p_tlpk_last_error)(const tlpk_handle *);
static const char *(*p_tlpk_backend_name)(void);
static const char *(*p_tlpk_system_name)(void);
static int (*p_tlpk_device_count)(void);

/* residual norms of test.jl:39-43 for a 1-based CSC matrix */
static void residuals(int64_t m, int64_t n, const int64_t *cp, const int64_t *ri, const double *v, const double *th,
                      const double *rp, const double *rd, const double *xp, const double *xd, const double *dx,
                      const double *dy, double *r1, double *r2) {
    double *rowacc = calloc((size_t)m, sizeof(double));
    *r2 = 0.0;
    for (int64_t j = 0; j < n; ++j) {
        double s = 0.0;
        for (int64_t p = cp[j] - 1; p < cp[j + 1] - 1; ++p) { rowacc[ri[p] - 1] += v[p] * dx[j]; s += v[p] * dy[ri[p] - 1]; }
        const double r = -dx[j] * (th[j] + rp[j]) + s - xd[j];
        if (fabs(r) > *r2) *r2 = fabs(r);
    }
    *r1 = 0.0;
    for (int64_t i = 0; i < m; ++i) { const double r = rowacc[i] + rd[i] * dy[i] - xp[i]; if (fabs(r) > *r1) *r1 = fabs(r); }
    free(rowacc);
}

static double lcg(uint64_t *s) { *s = *s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(*s >> 11) / 9007199254740992.0; }

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: abi_smoke <libtlpk.so> cpu|gpu\n"); return 2; }
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(tlpk_default_options); LOAD(tlpk_create); LOAD(tlpk_destroy); LOAD(tlpk_update); LOAD(tlpk_solve);
    LOAD(tlpk_info); LOAD(tlpk_get_perm); LOAD(tlpk_strerror); LOAD(tlpk_last_error); LOAD(tlpk_backend_name);
    LOAD(tlpk_system_name); LOAD(tlpk_device_count);
    const int gpu = strcmp(argv[2], "gpu") == 0;

    /* the reference fixture A = [1 0 1 0; 0 1 0 1], 1-based like Julia */
    const int64_t cp[5] = {1, 2, 3, 4, 5}, ri[4] = {1, 2, 1, 2};
    const double av[4] = {1.0, 1.0, 1.0, 1.0};
    tlpk_options opt;
    p_tlpk_default_options(&opt);
    CHECK(opt.struct_size == (int32_t)sizeof(tlpk_options) && opt.nranks == 1);
    opt.device = gpu ? 0 : -1;
    tlpk_handle *h = NULL;
    int rc = p_tlpk_create(&h, 2, 4, cp, ri, av, /*index_base=*/1, &opt);
    if (rc != TLPK_OK) { fprintf(stderr, "tlpk_create: %s\n", p_tlpk_strerror(rc)); return 1; }
    tlpk_stats st;
    CHECK(p_tlpk_info(h, &st) == TLPK_OK && st.m == 2 && st.n == 4 && st.nnzA == 4 && st.nnzS == 2 && st.nnzL == 2);
    int64_t perm[2];
    CHECK(p_tlpk_get_perm(h, perm) == TLPK_OK && perm[0] + perm[1] == 1);
    CHECK(strcmp(p_tlpk_backend_name(), "HIP (gfx950)") == 0);
    const double ones4[4] = {1, 1, 1, 1}, ones2[2] = {1, 1};
    double dx[4] = {9, 9, 9, 9}, dy[2] = {9, 9};
    if (!gpu) {
        CHECK(p_tlpk_update(h, ones4, ones4, ones2) == TLPK_NO_DEVICE);     /* no CPU fallback */
        CHECK(p_tlpk_solve(h, dx, dy, ones2, ones4) == TLPK_NO_DEVICE);
        CHECK(p_tlpk_update(NULL, ones4, ones4, ones2) == TLPK_BADARG);
        p_tlpk_destroy(h);
        /* wrong struct_size is rejected, no handle is returned */
        tlpk_options bad = opt; bad.struct_size = 4; h = (tlpk_handle *)1;
        CHECK(p_tlpk_create(&h, 2, 4, cp, ri, av, 1, &bad) == TLPK_BADARG && h == NULL);
        printf("abi_smoke cpu ok\n");
        return 0;
    }
    CHECK(p_tlpk_device_count() >= 1);
    CHECK(p_tlpk_solve(h, dx, dy, ones2, ones4) == TLPK_NOT_FACTORED);
    CHECK(p_tlpk_update(h, ones4, ones4, ones2) == TLPK_OK);                /* S = 2 I */
    CHECK(p_tlpk_solve(h, dx, dy, ones2, ones4) == TLPK_OK);
    CHECK(fabs(dy[0] - 1.0) < 1e-15 && fabs(dy[1] - 1.0) < 1e-15);
    for (int j = 0; j < 4; ++j) CHECK(fabs(dx[j]) < 1e-15);
    /* PosDefException contract (spd.jl:46-47; HSD/step.jl:35-49): error code, handle stays usable */
    const double negd[2] = {-10.0, 1.0};
    CHECK(p_tlpk_update(h, ones4, ones4, negd) == TLPK_NOT_POSDEF);
    CHECK(p_tlpk_info(h, &st) == TLPK_OK && st.fail_col >= 0);
    CHECK(p_tlpk_solve(h, dx, dy, ones2, ones4) == TLPK_NOT_FACTORED);
    CHECK(p_tlpk_update(h, ones4, ones4, ones2) == TLPK_OK);
    p_tlpk_destroy(h);

    /* a random 1-based LP matrix: m = 300, n = 700, 3 entries per column */
    const int64_t m = 300, n = 700, k = 3;
    int64_t *cp2 = malloc((size_t)(n + 1) * sizeof(int64_t)), *ri2 = malloc((size_t)(n * k) * sizeof(int64_t));
    double *v2 = malloc((size_t)(n * k) * sizeof(double));
    uint64_t s = 12345;
    for (int64_t j = 0; j < n; ++j) {
        cp2[j] = 1 + j * k;
        int64_t r0 = (int64_t)(lcg(&s) * (double)(m - k));
        for (int64_t t = 0; t < k; ++t) { ri2[j * k + t] = 1 + r0 + t; v2[j * k + t] = 2.0 * lcg(&s) - 1.0; }   /* sorted, distinct */
    }
    cp2[n] = 1 + n * k;
    double *th = malloc(n * sizeof(double)), *rp = malloc(n * sizeof(double)), *rd = malloc(m * sizeof(double));
    double *xp = malloc(m * sizeof(double)), *xd = malloc(n * sizeof(double)), *dx2 = malloc(n * sizeof(double)), *dy2 = malloc(m * sizeof(double));
    for (int64_t j = 0; j < n; ++j) { th[j] = pow(10.0, 6.0 * lcg(&s) - 3.0); rp[j] = 1e-4; xd[j] = 2.0 * lcg(&s) - 1.0; }
    for (int64_t i = 0; i < m; ++i) { rd[i] = 1e-4; xp[i] = 2.0 * lcg(&s) - 1.0; }
    CHECK(p_tlpk_create(&h, m, n, cp2, ri2, v2, 1, &opt) == TLPK_OK);
    CHECK(p_tlpk_update(h, th, rp, rd) == TLPK_OK);
    for (int64_t j = 0; j < n; ++j) th[j] = 1e30;              /* the caller may clobber its vectors right after update! (spd.jl:36-38) */
    CHECK(p_tlpk_solve(h, dx2, dy2, xp, xd) == TLPK_OK);
    s = 12345;                                                  /* regenerate th for the residual check */
    for (int64_t j = 0; j < n; ++j) { (void)lcg(&s); for (int64_t t = 0; t < k; ++t) (void)lcg(&s); }
    for (int64_t j = 0; j < n; ++j) { th[j] = pow(10.0, 6.0 * lcg(&s) - 3.0); (void)lcg(&s); }
    double r1, r2, dymax = 1.0, dxmax = 1.0;
    residuals(m, n, cp2, ri2, v2, th, rp, rd, xp, xd, dx2, dy2, &r1, &r2);
    for (int64_t i = 0; i < m; ++i) if (fabs(dy2[i]) > dymax) dymax = fabs(dy2[i]);
    for (int64_t j = 0; j < n; ++j) if (fabs(dx2[j]) > dxmax) dxmax = fabs(dx2[j]);
    printf("abi_smoke gpu: residuals %.3e %.3e (|dy| %.2e |dx| %.2e)\n", r1, r2, dymax, dxmax);
    CHECK(r1 <= 1e-8 * dymax && r2 <= 1e-8 * dxmax);
    /* dimension errors are return codes, never aborts */
    CHECK(p_tlpk_update(h, NULL, rp, rd) == TLPK_BADARG);
    p_tlpk_destroy(h);
    p_tlpk_destroy(NULL);
    printf("abi_smoke gpu ok\n");
    return 0;
}
