// ipm_kernels.hip -- device-resident interior-point vectors for the homogeneous self-dual loop
// (SURVEY.md 8(f)2-3): residuals + status quantities, construction of the Newton right-hand sides,
// recovery of the search direction, step lengths and the point update, all on vectors that never
// leave HBM.  The host (tlpk_ipm.cpp, tulip.jl_amd/hsd_device.py) exchanges only scalars.
//
// What is restated, kernel by kernel (citations into /root/reference):
//   k_ipm_res_cols / k_ipm_res_rows   src/IPM/HSD/HSD.jl:77-128 (residuals, norms, objectives) and the
//                                     quantities of the stopping tests, HSD.jl:136-196
//   k_ipm_theta                       src/IPM/HSD/step.jl:24-31   (theta_inv, regularisations)
//   k_ipm_hrhs, k_ipm_hdots           step.jl:56-76              (the h-system and h0)
//   k_ipm_targets                     step.jl:325-375            (Gondzio centrality targets)
//   k_ipm_newton_pre / _dots / _post  step.jl:198-266            (solve_newton_system) + step.jl:274-306 (max step)
//   k_ipm_advance                     step.jl:139-148            (point update)
//
// All HBM-bound elementwise / SpMV / reduction work: coalesced loads, one pass per vector, block
// reductions through LDS with a fixed tree, per-block partials combined in block order by a one-block
// finalize kernel (deterministic, no floating-point atomics).
#include <hip/hip_runtime.h>

#include "tlpk_ipm.hpp"

namespace tlpk {

constexpr int IPM_T = 256;

__device__ __forceinline__ double blk_sum(double v, double *sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = IPM_T / 2; s > 0; s >>= 1) { if (tid < s) sh[tid] += sh[tid + s]; __syncthreads(); }
    const double r = sh[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double blk_max(double v, double *sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = IPM_T / 2; s > 0; s >>= 1) { if (tid < s) sh[tid] = fmax(sh[tid], sh[tid + s]); __syncthreads(); }
    const double r = sh[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double blk_min(double v, double *sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = IPM_T / 2; s > 0; s >>= 1) { if (tid < s) sh[tid] = fmin(sh[tid], sh[tid + s]); __syncthreads(); }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// partials[block][slot]; slots [0, nsum) are sums, [nsum, nsum + nmax) maxima, the rest minima.  One wave per slot: lane l combines the blocks l, l + 64, ... in
// that order, then a fixed butterfly over the 64 lanes -- the same order in every run (bitwise deterministic), whatever the number of blocks.  (Until round 5 one
// THREAD per slot walked the 1024 blocks one dependent load at a time: 363 us per call, ten calls per interior-point iteration.)
__global__ __launch_bounds__(64 * IPM_SLOTS) void k_ipm_finalize(int nblocks, int nsum, int nmax, int nmin, const double *__restrict__ partials, double *__restrict__ out) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= nsum + nmax + nmin) return;                                     // (wave-uniform)
    const int op = (k < nsum) ? 0 : (k < nsum + nmax ? 1 : 2);
    double r = (op == 0) ? 0.0 : (op == 1 ? -INFINITY : INFINITY);
    for (int b = lane; b < nblocks; b += 64) {
        const double v = partials[(size_t)b * IPM_SLOTS + k];
        r = (op == 0) ? r + v : (op == 1 ? fmax(r, v) : fmin(r, v));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double v = __shfl_xor(r, off, 64);
        r = (op == 0) ? r + v : (op == 1 ? fmax(r, v) : fmin(r, v));
    }
    if (lane == 0) out[k] = r;
}

__global__ void k_ipm_init(IpmVecs v) {                                     // HSD.jl:238-247
    const i64 stride = (i64)gridDim.x * blockDim.x, t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (i64 j = t0; j < v.n; j += stride) { v.x[j] = 0.0; v.xl[j] = v.lflag[j]; v.xu[j] = v.uflag[j]; v.zl[j] = v.lflag[j]; v.zu[j] = v.uflag[j]; }
    for (i64 i = t0; i < v.m; i += stride) v.y[i] = 0.0;
}

// columns: rl, ru, rd + sums {c'x, lz'zl, uz'zu, xl'zl + xu'zu} + maxima {|rl|, |ru|, |rd|, |(x-xl) lflag|, |(x+xu) uflag|, |A'y + zl - zu|}
__global__ __launch_bounds__(IPM_T) void k_ipm_res_cols(IpmVecs v, double tau, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) {
        double aty = 0.0;
        for (i64 p = v.Ap[j]; p < v.Ap[j + 1]; ++p) aty += v.Ax[p] * v.y[v.Ai[p]];
        const double x = v.x[j], xl = v.xl[j], xu = v.xu[j], zl = v.zl[j], zu = v.zu[j], lf = v.lflag[j], uf = v.uflag[j];
        const double rl = (-x + xl + tau * v.lz[j]) * lf, ru = (-x - xu + tau * v.uz[j]) * uf;
        const double rd = tau * v.c[j] - aty + zu * uf - zl * lf;
        v.rl[j] = rl; v.ru[j] = ru; v.rd[j] = rd;
        s0 += v.c[j] * x; s1 += v.lz[j] * zl; s2 += v.uz[j] * zu; s3 += xl * zl + xu * zu;
        m0 = fmax(m0, fabs(rl)); m1 = fmax(m1, fabs(ru)); m2 = fmax(m2, fabs(rd));
        m3 = fmax(m3, fabs((x - xl) * lf)); m4 = fmax(m4, fabs((x + xu) * uf)); m5 = fmax(m5, fabs(aty + zl * lf - zu * uf));
    }
    double *P = partials + (size_t)blockIdx.x * IPM_SLOTS;
    double r;
    r = blk_sum(s0, sh); if (threadIdx.x == 0) P[0] = r;
    r = blk_sum(s1, sh); if (threadIdx.x == 0) P[1] = r;
    r = blk_sum(s2, sh); if (threadIdx.x == 0) P[2] = r;
    r = blk_sum(s3, sh); if (threadIdx.x == 0) P[3] = r;
    r = blk_max(m0, sh); if (threadIdx.x == 0) P[4] = r;
    r = blk_max(m1, sh); if (threadIdx.x == 0) P[5] = r;
    r = blk_max(m2, sh); if (threadIdx.x == 0) P[6] = r;
    r = blk_max(m3, sh); if (threadIdx.x == 0) P[7] = r;
    r = blk_max(m4, sh); if (threadIdx.x == 0) P[8] = r;
    r = blk_max(m5, sh); if (threadIdx.x == 0) P[9] = r;
}
// rows: rp + sum {b'y} + maxima {|rp|, |A x|}
__global__ __launch_bounds__(IPM_T) void k_ipm_res_rows(IpmVecs v, double tau, double *__restrict__ partials) {
    // 8 lanes per row, fixed shuffle tree inside the group (as k_rhs: one thread per row walked its ~9 entries one dependent gather at a time, 441 us on config C4)
    __shared__ double sh[IPM_T];
    double s0 = 0, m0 = 0, m1 = 0;
    const int lane = threadIdx.x & 7;
    const i64 stride = ((i64)gridDim.x * blockDim.x) >> 3;
    const i64 mround = (v.m + stride - 1) / stride * stride;                 // every group runs the same number of trips: the shuffles stay convergent
    for (i64 i = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < mround; i += stride) {
        const bool live = i < v.m;
        double ax = 0.0;
        if (live)
            for (i64 q = v.Tp[i] + lane; q < v.Tp[i + 1]; q += 8) ax += v.Tx[q] * v.x[v.Tj[q]];
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) ax += __shfl_down(ax, off, 8);
        if (!live || lane != 0) continue;
        const double rp = tau * v.b[i] - ax;
        v.rp[i] = rp;
        s0 += v.b[i] * v.y[i];
        if (v.row_skip && v.row_skip[i]) continue;                          // a shard's PARTIAL linking row: the host sums the shards' rows
        m0 = fmax(m0, fabs(rp)); m1 = fmax(m1, fabs(ax));
    }
    double *P = partials + (size_t)blockIdx.x * IPM_SLOTS;
    double r;
    r = blk_sum(s0, sh); if (threadIdx.x == 0) P[0] = r;
    r = blk_max(m0, sh); if (threadIdx.x == 0) P[1] = r;
    r = blk_max(m1, sh); if (threadIdx.x == 0) P[2] = r;
}

// theta_inv = zl/xl + zu/xu (exactly 0 for free variables), uniform regularisation vectors   step.jl:24-31
__global__ void k_ipm_theta(IpmVecs v, double *__restrict__ theta, double *__restrict__ regP, double *__restrict__ regD, double rP, double rD) {
    const i64 stride = (i64)gridDim.x * blockDim.x, t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (i64 j = t0; j < v.n; j += stride) {
        const double tl = (v.lflag[j] != 0.0) ? v.zl[j] / v.xl[j] : 0.0, tu = (v.uflag[j] != 0.0) ? v.zu[j] / v.xu[j] : 0.0;
        v.thl[j] = tl; v.thu[j] = tu; theta[j] = tl + tu; regP[j] = rP;
    }
    for (i64 i = t0; i < v.m; i += stride) regD[i] = rD;
}
__global__ void k_ipm_hrhs(IpmVecs v) {                                     // step.jl:61: xi_ = c - th_l lz - th_u uz
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) v.hxid[j] = v.c[j] - v.thl[j] * v.lz[j] - v.thu[j] * v.uz[j];
}
// h0 pieces (step.jl:69-76): sum_j lz^2 th_l + uz^2 th_u - (c + th_l lz + th_u uz) hx ; sum_i b hy
__global__ __launch_bounds__(IPM_T) void k_ipm_hdots(IpmVecs v, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double s0 = 0, s1 = 0;
    const i64 stride = (i64)gridDim.x * blockDim.x, t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (i64 j = t0; j < v.n; j += stride) {
        const double lz = v.lz[j], uz = v.uz[j], tl = v.thl[j], tu = v.thu[j];
        s0 += lz * (lz * tl) + uz * (uz * tu) - (v.c[j] + tl * lz + tu * uz) * v.hx[j];
    }
    for (i64 i = t0; i < v.m; i += stride) s1 += v.b[i] * v.hy[i];
    double *P = partials + (size_t)blockIdx.x * IPM_SLOTS;
    double r;
    r = blk_sum(s0, sh); if (threadIdx.x == 0) P[0] = r;
    r = blk_sum(s1, sh); if (threadIdx.x == 0) P[1] = r;
}

// Gondzio targets (step.jl:333-364): v = (x + a dx)(z + a dz) on bounded entries, mapped to the box [mu_l, mu_u];
// stored in xzl / xzu, sums returned (the host adds the tau-kappa term and forms delta).  HSD trials one step
// length (a_p = a_d); MPC separate primal and dual ones (MPC/step.jl:329-358)
__global__ __launch_bounds__(IPM_T) void k_ipm_targets(IpmVecs v, IpmDir D, double a_p, double a_d, double mu_l, double mu_u, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double s0 = 0, s1 = 0;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) {
        double vl = ((v.xl[j] + a_p * D.xl[j]) * (v.zl[j] + a_d * D.zl[j])) * v.lflag[j];
        double vu = ((v.xu[j] + a_p * D.xu[j]) * (v.zu[j] + a_d * D.zu[j])) * v.uflag[j];
        if (v.lflag[j] != 0.0) vl = (vl < mu_l) ? mu_l - vl : ((vl > mu_u) ? mu_u - vl : 0.0);
        if (v.uflag[j] != 0.0) vu = (vu < mu_l) ? mu_l - vu : ((vu > mu_u) ? mu_u - vu : 0.0);
        v.xzl[j] = vl; v.xzu[j] = vu;
        s0 += vl; s1 += vu;
    }
    double *P = partials + (size_t)blockIdx.x * IPM_SLOTS;
    double r;
    r = blk_sum(s0, sh); if (threadIdx.x == 0) P[0] = r;
    r = blk_sum(s1, sh); if (threadIdx.x == 0) P[1] = r;
}

// Right-hand sides of one Newton system (step.jl:198-222).  mode 0: predictor (xi = residuals, complementarity
// -x z); mode 1: corrector (eta-scaled residuals, -x z + gamma mu - dx dz with the predictor direction D);
// mode 2: centrality corrector (zero residuals, targets in xzl / xzu minus delta).
__global__ __launch_bounds__(IPM_T) void k_ipm_newton_pre(IpmVecs v, IpmDir D, int mode, double eta, double gmu, double delta, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const i64 stride = (i64)gridDim.x * blockDim.x, t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (i64 j = t0; j < v.n; j += stride) {
        const double lf = v.lflag[j], uf = v.uflag[j], xl = v.xl[j], xu = v.xu[j], zl = v.zl[j], zu = v.zu[j];
        double xil, xiu, xd, xzl, xzu;
        if (mode == 0) { xil = v.rl[j]; xiu = v.ru[j]; xd = v.rd[j]; xzl = -(xl * zl) * lf; xzu = -(xu * zu) * uf; }
        else if (mode == 1) {
            xil = eta * v.rl[j]; xiu = eta * v.ru[j]; xd = eta * v.rd[j];
            xzl = (-xl * zl + gmu - D.xl[j] * D.zl[j]) * lf; xzu = (-xu * zu + gmu - D.xu[j] * D.zu[j]) * uf;
        } else { xil = 0.0; xiu = 0.0; xd = 0.0; xzl = v.xzl[j] - delta; xzu = v.xzu[j] - delta; }
        v.xil[j] = xil; v.xiu[j] = xiu; v.xzl[j] = xzl; v.xzu[j] = xzu;
        const double tl = (lf != 0.0) ? (xzl + zl * xil) / xl : 0.0, tu = (uf != 0.0) ? (xzu - zu * xiu) / xu : 0.0;
        v.xid[j] = xd - tl + tu;                                            // step.jl:214
        const double ixl = (lf != 0.0) ? xzl / xl : 0.0, ixu = (uf != 0.0) ? xzu / xu : 0.0;
        s0 += ixl * v.lz[j]; s1 += ixu * v.uz[j]; s2 += (v.thl[j] * xil) * v.lz[j]; s3 += (v.thu[j] * xiu) * v.uz[j];
    }
    for (i64 i = t0; i < v.m; i += stride) v.xip[i] = (mode == 0) ? v.rp[i] : (mode == 1 ? eta * v.rp[i] : 0.0);
    double *P = partials + (size_t)blockIdx.x * IPM_SLOTS;
    double r;
    r = blk_sum(s0, sh); if (threadIdx.x == 0) P[0] = r;
    r = blk_sum(s1, sh); if (threadIdx.x == 0) P[1] = r;
    r = blk_sum(s2, sh); if (threadIdx.x == 0) P[2] = r;
    r = blk_sum(s3, sh); if (threadIdx.x == 0) P[3] = r;
}
// (c + th_l lz + th_u uz)' dx ; b' dy     (step.jl:240-246)
__global__ __launch_bounds__(IPM_T) void k_ipm_newton_dots(IpmVecs v, IpmDir D, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double s0 = 0, s1 = 0;
    const i64 stride = (i64)gridDim.x * blockDim.x, t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (i64 j = t0; j < v.n; j += stride) s0 += (v.c[j] + v.thl[j] * v.lz[j] + v.thu[j] * v.uz[j]) * D.x[j];
    for (i64 i = t0; i < v.m; i += stride) s1 += v.b[i] * D.y[i];
    double *P = partials + (size_t)blockIdx.x * IPM_SLOTS;
    double r;
    r = blk_sum(s0, sh); if (threadIdx.x == 0) P[4] = r;                     // slots 4, 5: the pre kernel's sums stay in 0..3
    r = blk_sum(s1, sh); if (threadIdx.x == 0) P[5] = r;
}
// Direction recovery (step.jl:248-263), optional accumulation Dc += D (centrality corrector, step.jl:377-386) and
// the largest step to the boundary (step.jl:274-306) of the resulting direction.
__global__ __launch_bounds__(IPM_T) void k_ipm_newton_post(IpmVecs v, IpmDir D, IpmDir Add, int add, double dtau, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double amin_p = __builtin_inf(), amin_d = __builtin_inf();               // primal (xl, xu) and dual (zl, zu) sides
    const i64 stride = (i64)gridDim.x * blockDim.x, t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (i64 j = t0; j < v.n; j += stride) {
        const double lf = v.lflag[j], uf = v.uflag[j];
        double dx = D.x[j] + dtau * v.hx[j];
        double dxl = (-v.xil[j] + dx - dtau * v.lz[j]) * lf, dxu = (v.xiu[j] - dx + dtau * v.uz[j]) * uf;
        double dzl = (lf != 0.0) ? (v.xzl[j] - v.zl[j] * dxl) / v.xl[j] : 0.0, dzu = (uf != 0.0) ? (v.xzu[j] - v.zu[j] * dxu) / v.xu[j] : 0.0;
        if (add) { dx += Add.x[j]; dxl += Add.xl[j]; dxu += Add.xu[j]; dzl += Add.zl[j]; dzu += Add.zu[j]; }
        D.x[j] = dx; D.xl[j] = dxl; D.xu[j] = dxu; D.zl[j] = dzl; D.zu[j] = dzu;
        if (dxl < 0.0) amin_p = fmin(amin_p, -v.xl[j] / dxl);
        if (dxu < 0.0) amin_p = fmin(amin_p, -v.xu[j] / dxu);
        if (dzl < 0.0) amin_d = fmin(amin_d, -v.zl[j] / dzl);
        if (dzu < 0.0) amin_d = fmin(amin_d, -v.zu[j] / dzu);
    }
    for (i64 i = t0; i < v.m; i += stride) { double dy = D.y[i] + dtau * v.hy[i]; if (add) dy += Add.y[i]; D.y[i] = dy; }
    double r = blk_min(amin_p, sh);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * IPM_SLOTS + 0] = r;
    r = blk_min(amin_d, sh);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * IPM_SLOTS + 1] = r;
}
// pt += alpha D (step.jl:139-148); returns xl'zl + xu'zu of the new point for mu (point.jl:45-48)
// (MPC/step.jl:112-123: primal side by alpha_p, dual side by alpha_d; HSD passes the same value twice)
__global__ __launch_bounds__(IPM_T) void k_ipm_advance(IpmVecs v, IpmDir D, double alpha, double alpha_d, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double s0 = 0;
    const i64 stride = (i64)gridDim.x * blockDim.x, t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (i64 j = t0; j < v.n; j += stride) {
        v.x[j] += alpha * D.x[j];
        const double xl = v.xl[j] + alpha * D.xl[j], xu = v.xu[j] + alpha * D.xu[j], zl = v.zl[j] + alpha_d * D.zl[j], zu = v.zu[j] + alpha_d * D.zu[j];
        v.xl[j] = xl; v.xu[j] = xu; v.zl[j] = zl; v.zu[j] = zu;
        s0 += xl * zl + xu * zu;
    }
    for (i64 i = t0; i < v.m; i += stride) v.y[i] += alpha_d * D.y[i];
    const double r = blk_sum(s0, sh);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * IPM_SLOTS + 0] = r;
}


// ---------------------------------------------------------------------------------------------
// Mehrotra predictor-corrector (the non-homogeneous algorithm, /root/reference/src/IPM/MPC): what the kernels
// above do not already cover with tau = 1, dtau = 0.
//   k_mpc_fill            MPC.jl:359             theta_inv = 0, regP = 1, regD = 1e-6 of the starting-point system
//   k_mpc_start1 .. 4     MPC.jl:365-405         shifts to positive coordinates, balanced complementarity products
//   k_mpc_gap             MPC/step.jl:246-258, 290-296   complementarity after a trial step (mu_aff, corrector targets)
// ---------------------------------------------------------------------------------------------
__global__ void k_mpc_fill(IpmVecs v, double *__restrict__ theta, double *__restrict__ regP, double *__restrict__ regD) {
    const i64 stride = (i64)gridDim.x * blockDim.x, t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    for (i64 j = t0; j < v.n; j += stride) { theta[j] = 0.0; regP[j] = 1.0; v.xid[j] = 0.0; v.hx[j] = 0.0; }
    for (i64 i = t0; i < v.m; i += stride) { regD[i] = 1e-6; v.xip[i] = 0.0; v.hy[i] = 0.0; }
}
// minima of (x - l) lflag and (u - x) uflag (entries without the bound contribute 0, as `false * Inf` does there)
__global__ __launch_bounds__(IPM_T) void k_mpc_start1(IpmVecs v, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double m0 = 0.0, m1 = 0.0;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) {
        if (v.lflag[j] != 0.0) m0 = fmin(m0, v.x[j] - v.lz[j]);
        if (v.uflag[j] != 0.0) m1 = fmin(m1, v.uz[j] - v.x[j]);
    }
    double r = blk_min(m0, sh); if (threadIdx.x == 0) partials[(size_t)blockIdx.x * IPM_SLOTS + 0] = r;
    r = blk_min(m1, sh); if (threadIdx.x == 0) partials[(size_t)blockIdx.x * IPM_SLOTS + 1] = r;
}
// xl, xu shifted by dxs; z = c - A'y split over the finite bounds; minima of zl, zu
__global__ __launch_bounds__(IPM_T) void k_mpc_start2(IpmVecs v, double dxs, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double m0 = 0.0, m1 = 0.0;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) {
        const double lf = v.lflag[j], uf = v.uflag[j], x = v.x[j];
        v.xl[j] = (lf != 0.0) ? (x - v.lz[j]) + dxs : 0.0;
        v.xu[j] = (uf != 0.0) ? (v.uz[j] - x) + dxs : 0.0;
        double aty = 0.0;
        for (i64 p = v.Ap[j]; p < v.Ap[j + 1]; ++p) aty += v.Ax[p] * v.y[v.Ai[p]];
        const double z = v.c[j] - aty, nb = lf + uf;
        const double zl = (lf != 0.0) ? z / nb : 0.0, zu = (uf != 0.0) ? -z / nb : 0.0;
        v.zl[j] = zl; v.zu[j] = zu;
        m0 = fmin(m0, zl); m1 = fmin(m1, zu);
    }
    double r = blk_min(m0, sh); if (threadIdx.x == 0) partials[(size_t)blockIdx.x * IPM_SLOTS + 0] = r;
    r = blk_min(m1, sh); if (threadIdx.x == 0) partials[(size_t)blockIdx.x * IPM_SLOTS + 1] = r;
}
// zl, zu shifted by dzs; sums {xl'zl + xu'zu, sum zl + sum zu, sum xl + sum xu}
__global__ __launch_bounds__(IPM_T) void k_mpc_start3(IpmVecs v, double dzs, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double s0 = 0, s1 = 0, s2 = 0;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) {
        const double zl = (v.lflag[j] != 0.0) ? v.zl[j] + dzs : v.zl[j], zu = (v.uflag[j] != 0.0) ? v.zu[j] + dzs : v.zu[j];
        v.zl[j] = zl; v.zu[j] = zu;
        const double xl = v.xl[j], xu = v.xu[j];
        s0 += xl * zl + xu * zu; s1 += zl + zu; s2 += xl + xu;
    }
    double *P = partials + (size_t)blockIdx.x * IPM_SLOTS;
    double r = blk_sum(s0, sh); if (threadIdx.x == 0) P[0] = r;
    r = blk_sum(s1, sh); if (threadIdx.x == 0) P[1] = r;
    r = blk_sum(s2, sh); if (threadIdx.x == 0) P[2] = r;
}
// balanced products: xl, xu += ddx, zl, zu += ddz on the finite bounds; returns xl'zl + xu'zu
__global__ __launch_bounds__(IPM_T) void k_mpc_start4(IpmVecs v, double ddx, double ddz, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double s0 = 0;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) {
        double xl = v.xl[j], xu = v.xu[j], zl = v.zl[j], zu = v.zu[j];
        if (v.lflag[j] != 0.0) { xl += ddx; zl += ddz; }
        if (v.uflag[j] != 0.0) { xu += ddx; zu += ddz; }
        v.xl[j] = xl; v.xu[j] = xu; v.zl[j] = zl; v.zu[j] = zu;
        s0 += xl * zl + xu * zu;
    }
    const double r = blk_sum(s0, sh);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * IPM_SLOTS + 0] = r;
}
// {((xl + ap dxl) lflag)'(zl + ad dzl) + ((xu + ap dxu) uflag)'(zu + ad dzu), xl'zl + xu'zu}
__global__ __launch_bounds__(IPM_T) void k_mpc_gap(IpmVecs v, IpmDir D, double ap, double ad, double *__restrict__ partials) {
    __shared__ double sh[IPM_T];
    double s0 = 0, s1 = 0;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < v.n; j += stride) {
        const double xl = v.xl[j], xu = v.xu[j], zl = v.zl[j], zu = v.zu[j];
        s0 += ((xl + ap * D.xl[j]) * v.lflag[j]) * (zl + ad * D.zl[j]) + ((xu + ap * D.xu[j]) * v.uflag[j]) * (zu + ad * D.zu[j]);
        s1 += xl * zl + xu * zu;
    }
    double *P = partials + (size_t)blockIdx.x * IPM_SLOTS;
    double r = blk_sum(s0, sh); if (threadIdx.x == 0) P[0] = r;
    r = blk_sum(s1, sh); if (threadIdx.x == 0) P[1] = r;
}

// ---------------------------------------------------------------------------------------------
static inline int ipm_blocks(i64 len) { return (int)std::max<i64>(1, std::min<i64>(IPM_BLOCKS, (len + IPM_T - 1) / IPM_T)); }

void ipm_launch_init(hipStream_t st, const IpmVecs &v) { hipLaunchKernelGGL(k_ipm_init, dim3(ipm_blocks(std::max(v.n, v.m))), dim3(IPM_T), 0, st, v); }
void ipm_launch_finalize(hipStream_t st, int nblocks, int nsum, int nmax, int nmin, const double *partials, double *out) {
    hipLaunchKernelGGL(k_ipm_finalize, dim3(1), dim3(64 * (nsum + nmax + nmin)), 0, st, nblocks, nsum, nmax, nmin, partials, out);
}
int ipm_launch_res_cols(hipStream_t st, const IpmVecs &v, double tau, double *partials) { const int nb = ipm_blocks(v.n); hipLaunchKernelGGL(k_ipm_res_cols, dim3(nb), dim3(IPM_T), 0, st, v, tau, partials); return nb; }
int ipm_launch_res_rows(hipStream_t st, const IpmVecs &v, double tau, double *partials) { const int nb = ipm_blocks(v.m); hipLaunchKernelGGL(k_ipm_res_rows, dim3(nb), dim3(IPM_T), 0, st, v, tau, partials); return nb; }
void ipm_launch_theta(hipStream_t st, const IpmVecs &v, double *theta, double *regP, double *regD, double rP, double rD) {
    hipLaunchKernelGGL(k_ipm_theta, dim3(ipm_blocks(std::max(v.n, v.m))), dim3(IPM_T), 0, st, v, theta, regP, regD, rP, rD);
}
void ipm_launch_hrhs(hipStream_t st, const IpmVecs &v) { hipLaunchKernelGGL(k_ipm_hrhs, dim3(ipm_blocks(v.n)), dim3(IPM_T), 0, st, v); }
int ipm_launch_hdots(hipStream_t st, const IpmVecs &v, double *partials) { const int nb = ipm_blocks(std::max(v.n, v.m)); hipLaunchKernelGGL(k_ipm_hdots, dim3(nb), dim3(IPM_T), 0, st, v, partials); return nb; }
int ipm_launch_targets(hipStream_t st, const IpmVecs &v, const IpmDir &D, double a_p, double a_d, double mu_l, double mu_u, double *partials) {
    const int nb = ipm_blocks(v.n); hipLaunchKernelGGL(k_ipm_targets, dim3(nb), dim3(IPM_T), 0, st, v, D, a_p, a_d, mu_l, mu_u, partials); return nb;
}
int ipm_launch_newton_pre(hipStream_t st, const IpmVecs &v, const IpmDir &D, int mode, double eta, double gmu, double delta, double *partials) {
    const int nb = ipm_blocks(std::max(v.n, v.m)); hipLaunchKernelGGL(k_ipm_newton_pre, dim3(nb), dim3(IPM_T), 0, st, v, D, mode, eta, gmu, delta, partials); return nb;
}
void ipm_launch_newton_dots(hipStream_t st, const IpmVecs &v, const IpmDir &D, int nblocks, double *partials) {
    hipLaunchKernelGGL(k_ipm_newton_dots, dim3(nblocks), dim3(IPM_T), 0, st, v, D, partials);
}
int ipm_launch_newton_post(hipStream_t st, const IpmVecs &v, const IpmDir &D, const IpmDir &Add, int add, double dtau, double *partials) {
    const int nb = ipm_blocks(std::max(v.n, v.m)); hipLaunchKernelGGL(k_ipm_newton_post, dim3(nb), dim3(IPM_T), 0, st, v, D, Add, add, dtau, partials); return nb;
}
int ipm_launch_advance(hipStream_t st, const IpmVecs &v, const IpmDir &D, double alpha_p, double alpha_d, double *partials) {
    const int nb = ipm_blocks(std::max(v.n, v.m)); hipLaunchKernelGGL(k_ipm_advance, dim3(nb), dim3(IPM_T), 0, st, v, D, alpha_p, alpha_d, partials); return nb;
}

void mpc_launch_fill(hipStream_t st, const IpmVecs &v, double *theta, double *regP, double *regD) {
    hipLaunchKernelGGL(k_mpc_fill, dim3(ipm_blocks(std::max(v.n, v.m))), dim3(IPM_T), 0, st, v, theta, regP, regD);
}
int mpc_launch_start(hipStream_t st, const IpmVecs &v, int stage, double a, double b, double *partials) {
    const int nb = ipm_blocks(v.n);
    if (stage == 1) hipLaunchKernelGGL(k_mpc_start1, dim3(nb), dim3(IPM_T), 0, st, v, partials);
    else if (stage == 2) hipLaunchKernelGGL(k_mpc_start2, dim3(nb), dim3(IPM_T), 0, st, v, a, partials);
    else if (stage == 3) hipLaunchKernelGGL(k_mpc_start3, dim3(nb), dim3(IPM_T), 0, st, v, a, partials);
    else hipLaunchKernelGGL(k_mpc_start4, dim3(nb), dim3(IPM_T), 0, st, v, a, b, partials);
    return nb;
}
int mpc_launch_gap(hipStream_t st, const IpmVecs &v, const IpmDir &D, double ap, double ad, double *partials) {
    const int nb = ipm_blocks(v.n); hipLaunchKernelGGL(k_mpc_gap, dim3(nb), dim3(IPM_T), 0, st, v, D, ap, ad, partials); return nb;
}

}  // namespace tlpk
