// mfma_f64_peak.hip -- micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950.
// The guide (/opt/skills/guides/MI355X_MICROARCH.md) does not tabulate the fp64 MFMA peak; the
// datasheet figure is 78.6 TFLOP/s (= 32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz).  This measures
// the ceiling our panel-update kernel is priced against: TFLOP/s from HIP events, cycles per
// MFMA from s_memtime (shader clock) inside the kernel, and the effective clock = cycles / time.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak && tools/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_peak(double *out, long long *cyc, int iters, double a0, double b0) {
    v4f64 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
void run(int blocks_per_cu, int iters, int threads = 256, int ncu = 256) {
    const int grid = ncu * blocks_per_cu;
    double *out; long long *cyc, hc = 0;
    hipMalloc(&out, sizeof(double) * grid * threads);
    hipMalloc(&cyc, sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_peak<NACC>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters, 1.0, 1e-9);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_peak<NACC>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters, 1.0, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipMemcpy(&hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    const int waves_per_block = threads / 64;
    const double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * waves_per_block * grid;
    const double waves_per_simd = (double)waves_per_block * blocks_per_cu / 4.0;
    printf("nacc=%2d waves/SIMD=%.2f CUs=%3d: %8.3f ms %7.2f TFLOP/s | s_memtime ticks/MFMA (one wave) %.1f | tick rate %.0f MHz\n",
           NACC, waves_per_simd, ncu, best, flops / best / 1e9, (double)hc / ((double)NACC * iters),
           (double)hc / (best * 1e-3) / 1e6);
    hipFree(out); hipFree(cyc);
}

int main() {
    printf("-- one CU worth of work (no power limit): --\n");
    run<4>(1, 20000, 256, 1);
    run<4>(2, 20000, 256, 1);
    run<4>(1, 20000, 64, 1);
    printf("-- full chip: --\n");
    run<1>(1, 20000);
    run<2>(1, 20000);
    run<4>(1, 20000);
    run<16>(1, 10000);
    run<4>(2, 20000);
    run<8>(2, 10000);
    run<16>(2, 10000);
    run<4>(4, 10000);
    run<4>(8, 5000);
    run<4>(1, 20000, 256, 128);
    run<4>(1, 20000, 256, 64);
    return 0;
}
