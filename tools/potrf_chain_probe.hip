// What does one column step of the register-resident 64 x 64 factorisation (potrf_block in kernels.hip) cost, piece by piece?
// One workgroup of 256 threads runs N dependent "column steps" of the same shape: (a) pivot column + inverse row through LDS,
// one barrier, ~18 broadcast LDS reads; (b) reciprocal square root + two Newton steps + corrected square root; (c) 17
// multiply-adds per thread.  Variants drop pieces; cycles per step from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o tools/potrf_chain_probe tools/potrf_chain_probe.hip && tools/potrf_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, int nsteps, double seed) {
    __shared__ double colbuf[2][64], rowbuf[2][64];
    const int tid = threadIdx.x, r = tid & 63, cg = tid >> 6;
    double av[16], wv[16];
    for (int q = 0; q < 16; ++q) { av[q] = seed + 0.001 * (r + 4 * q + cg); wv[q] = (r == cg + 4 * q) ? 1.0 : 0.0; }
    __syncthreads();
    const long long t0 = clock64();
    for (int rep = 0; rep < nsteps / 64; ++rep) {
#pragma unroll
    for (int jq = 0; jq < 16; ++jq) {
#pragma unroll 1
    for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * jq + jj, pb = j & 1;
        double d, crj, cv[16], rv[16];
        if (V != 2 && V != 4) {
            if (cg == jj) colbuf[pb][r] = av[jq];
            if (r == j) {
#pragma unroll
                for (int q = 0; q <= jq; ++q) rowbuf[pb][cg + 4 * q] = wv[q];
            }
            __syncthreads();
            d = colbuf[pb][j]; crj = colbuf[pb][r];
            if (V == 3) { av[jq] += d * 1e-9 + crj * 1e-12; continue; }
#pragma unroll
            for (int q = jq; q < 16; ++q) cv[q] = colbuf[pb][cg + 4 * q];
#pragma unroll
            for (int q = 0; q <= jq; ++q) rv[q] = rowbuf[pb][cg + 4 * q];
        } else {
            d = av[jq] + 2.0; crj = av[(jq + 1) & 15];
#pragma unroll
            for (int q = 0; q < 16; ++q) { cv[q] = av[(q + 1) & 15] * 1e-3; rv[q] = wv[(q + 1) & 15] * 1e-3; }
        }
        d = fabs(d) + 1.0;
        double isq, sq, inv2;
        if (V == 1 || V == 4) { isq = d * 0.25; sq = d * 0.5; inv2 = isq * isq; }
        else {
            isq = __builtin_amdgcn_rsq(d);
            isq = isq * (1.5 - 0.5 * d * isq * isq);
            isq = isq * (1.5 - 0.5 * d * isq * isq);
            sq = d * isq;
            sq = fma(0.5 * isq, fma(-sq, sq, d), sq);
            inv2 = isq * isq;
        }
        const double arj = (r > j) ? crj * inv2 * 1e-6 : 0.0;
#pragma unroll
        for (int q = jq + 1; q < 16; ++q) av[q] = fma(-arj, cv[q], av[q]);
        av[jq] = fma(-((cg > jj) ? arj : 0.0), cv[jq], av[jq]);
#pragma unroll
        for (int q = 0; q <= jq; ++q) wv[q] = fma(-arj, rv[q], wv[q]);
        if (cg == jj) av[jq] = (r == j) ? sq : ((r > j) ? av[jq] * (1.0 + isq * 1e-9) : av[jq]);
    }
    }
    }
    const long long t1 = clock64();
    double acc = 0; for (int q = 0; q < 16; ++q) acc += av[q] + wv[q];
    out[tid] = acc;
    if (tid == 0) cyc[0] = t1 - t0;
}
template <int V> void run(const char *name, double *d_out, long long *d_cyc) {
    const int N = 64 * 64;
    long long c = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<V>, dim3(1), dim3(256), 0, 0, d_out, d_cyc, N, 1.0 + rep);
        hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    }
    printf("%-62s %7.1f clock64 ticks / step\n", name, (double)c / N);
}
int main() {
    double *d_out; long long *d_cyc;
    hipMalloc(&d_out, 256 * 8); hipMalloc(&d_cyc, 8);
    run<0>("full step (LDS hand-over + barrier, rsq chain, 17 FMAs)", d_out, d_cyc);
    run<1>("without the reciprocal-square-root chain", d_out, d_cyc);
    run<2>("without LDS / barrier (operands from registers)", d_out, d_cyc);
    run<3>("LDS write + barrier + two dependent LDS reads only", d_out, d_cyc);
    run<4>("FMAs only", d_out, d_cyc);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    printf("(clock64 = s_memtime; wall clock rate attribute: %d kHz)\n", khz);
    return 0;
}
