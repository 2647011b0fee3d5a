// tlpk_ipm.hpp -- device-resident interior-point state: views shared by
// ipm_kernels.hip and tlpk_ipm.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "tlpk_host.hpp"

namespace tlpk {

constexpr int IPM_SLOTS = 16;      // reduction results per kernel
constexpr int IPM_BLOCKS = 1024;   // workgroups of the grid-stride kernels (per-block partials, fixed combination order)

// search direction (or any vector of the iterate's shape)
struct IpmDir { double *x, *xl, *xu, *zl, *zu /* n */, *y /* m */; };

// everything the kernels read; passed by value
struct IpmVecs {
    i64 m, n;
    const i64 *Ap; const i32 *Ai; const double *Ax;       // A, CSC (the handle's copy)
    const i64 *Tp; const i32 *Tj; const double *Tx;       // A, CSR
    const double *b, *c, *lz, *uz, *lflag, *uflag;        // problem data: l .* lflag, u .* uflag, flags as 0 / 1
    double *x, *xl, *xu, *zl, *zu, *y;                    // the iterate (point.jl)
    double *rp, *rl, *ru, *rd;                            // residuals (HSD.jl:83-110)
    double *thl, *thu;                                    // zl ./ xl, zu ./ xu on bounded entries
    double *hx, *hy;                                      // solution of the h-system (step.jl:56-66)
    double *hxid;                                         // its right-hand side c - th_l lz - th_u uz (own vector: the h-system and the
                                                          // predictor are solved as a pair, tlpk_ipm_hsolve_newton)
    double *xil, *xiu, *xzl, *xzu, *xid, *xip;            // right-hand sides of the current Newton system
    const char *row_skip;                                 // shard of a multi-device handle: 1 on the linking rows (partial sums there: no part in the maxima); else nullptr
};

void ipm_launch_init(hipStream_t st, const IpmVecs &v);
void ipm_launch_finalize(hipStream_t st, int nblocks, int nsum, int nmax, int nmin, const double *partials, double *out);
int ipm_launch_res_cols(hipStream_t st, const IpmVecs &v, double tau, double *partials);
int ipm_launch_res_rows(hipStream_t st, const IpmVecs &v, double tau, double *partials);
void ipm_launch_theta(hipStream_t st, const IpmVecs &v, double *theta, double *regP, double *regD, double rP, double rD);
void ipm_launch_hrhs(hipStream_t st, const IpmVecs &v);
int ipm_launch_hdots(hipStream_t st, const IpmVecs &v, double *partials);
int ipm_launch_targets(hipStream_t st, const IpmVecs &v, const IpmDir &D, double a_p, double a_d, double mu_l, double mu_u, double *partials);
int ipm_launch_newton_pre(hipStream_t st, const IpmVecs &v, const IpmDir &D, int mode, double eta, double gmu, double delta, double *partials);
void ipm_launch_newton_dots(hipStream_t st, const IpmVecs &v, const IpmDir &D, int nblocks, double *partials);
int ipm_launch_newton_post(hipStream_t st, const IpmVecs &v, const IpmDir &D, const IpmDir &Add, int add, double dtau, double *partials);
int ipm_launch_advance(hipStream_t st, const IpmVecs &v, const IpmDir &D, double alpha_p, double alpha_d, double *partials);
void mpc_launch_fill(hipStream_t st, const IpmVecs &v, double *theta, double *regP, double *regD);
int mpc_launch_start(hipStream_t st, const IpmVecs &v, int stage, double a, double b, double *partials);
int mpc_launch_gap(hipStream_t st, const IpmVecs &v, const IpmDir &D, double ap, double ad, double *partials);

}  // namespace tlpk
