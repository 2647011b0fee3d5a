// mfma_f64_variants.hip -- which fp64 matrix instruction / register class / occupancy reaches
// the highest sustained rate on gfx950?  (companion of mfma_f64_peak.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

// MODE 0: 16x16x4, accumulators in AGPRs (builtin, compiler's choice)
// MODE 1: 16x16x4, accumulators forced into VGPRs (inline asm)
// MODE 2: 4x4x4_4b (builtin)
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k_var(double *out, int iters, double a0, double b0) {
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    double s = 0;
    if constexpr (MODE == 0) {
        v4f64 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (MODE == 1) {
        v4f64 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(int blocks_per_cu, int iters, const char *name) {
    const int ncu = 256, grid = ncu * blocks_per_cu;
    double *out;
    hipMalloc(&out, sizeof(double) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_var<MODE, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-9);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_var<MODE, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double per = (MODE == 2) ? 2.0 * 4 * 4 * 4 * 4 : 2.0 * 16 * 16 * 4;
    const double flops = per * (double)NACC * iters * 4.0 * grid;
    printf("%-22s nacc=%2d waves/SIMD=%d : %8.3f ms %7.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", name, NACC, blocks_per_cu,
           best, flops / best / 1e9, best * 1e-3 * 2.4e9 / ((double)NACC * iters * blocks_per_cu));
    hipFree(out);
}

int main() {
    run<0, 8>(1, 10000, "16x16x4 agpr");
    run<1, 8>(1, 10000, "16x16x4 vgpr");
    run<0, 8>(2, 10000, "16x16x4 agpr");
    run<1, 8>(2, 10000, "16x16x4 vgpr");
    run<1, 4>(3, 10000, "16x16x4 vgpr");
    run<1, 4>(4, 10000, "16x16x4 vgpr");
    run<2, 8>(1, 40000, "4x4x4_4b");
    run<2, 8>(2, 40000, "4x4x4_4b");
    run<2, 8>(4, 40000, "4x4x4_4b");
    run<2, 16>(8, 20000, "4x4x4_4b");
    return 0;
}
