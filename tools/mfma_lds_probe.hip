// mfma_lds_probe.hip -- what does the inner loop of k_update cost beyond the matrix pipe?
// One workgroup = 4 waves, each a 64 x 64 register tile (16 accumulators of v_mfma_f64_16x16x4_f64);
// per K step of 4 every wave reads 4 + 4 operand fragments from LDS (layout of k_update: row stride
// 144 doubles).  Modes isolate: MFMA only / + LDS reads that feed nothing / reads feeding the MFMAs
// without and with a one-step software pipeline / 8 vs 16 accumulators.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_lds_probe.hip -o tools/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int LD = 144, KT = 16;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_probe(double *out, int rounds, double seed, const double *gsrc, long gstride) {
    __shared__ double As[2][KT * LD], Bs[2][KT * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lk = lane >> 4;
    for (int i = tid; i < 2 * KT * LD; i += 256) { (&As[0][0])[i] = seed + i * 1e-9; (&Bs[0][0])[i] = seed - i * 1e-9; }
    __syncthreads();
    v4f64 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4f64){0, 0, 0, 0};
    const double *At0 = As[0] + wr * 64 + lr + lk * LD, *Bt0 = Bs[0] + wc * 64 + lr + lk * LD;
    double av[4], bv[4], an[4], bn[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { av[a] = Bt0[a * 16]; bv[a] = At0[a * 16]; }
    auto mm = [&](double (&x)[4], double (&y)[4]) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[a][b]) : "v"(x[a]), "v"(y[b]));
    };
    double pa[8], pb[8], pa2[8], pb2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { pa[u] = seed; pb[u] = seed; pa2[u] = seed; pb2[u] = seed; }
    const int sr = tid & 127, sk0 = tid >> 7;
    const double *ga = gsrc + (size_t)blockIdx.x * 128 + sr, *gb = ga + 64 * 128;
    auto round = [&](auto PAR, int rd) {
        constexpr int par = decltype(PAR)::value;
        const double *At = At0 + par * KT * LD, *Bt = Bt0 + par * KT * LD;
        if constexpr (MODE >= 5 && MODE != 7) {         // slab rd+1 -> the other LDS buffer
            double *Aw = As[par ^ 1], *Bw = Bs[par ^ 1];
#pragma unroll
            for (int u = 0; u < 8; ++u) { Aw[(sk0 + 2 * u) * LD + sr] = pa[u]; Bw[(sk0 + 2 * u) * LD + sr] = pb[u]; }
        }
        if constexpr (MODE == 6) {                      // slab rd+2 from global memory (column stride gstride doubles)
#pragma unroll
            for (int u = 0; u < 8; ++u) { pa[u] = ga[(size_t)(sk0 + 2 * u) * gstride]; pb[u] = gb[(size_t)(sk0 + 2 * u) * gstride]; }
            ga += 16 * gstride; gb += 16 * gstride;
        }
        if constexpr (MODE == 7) {                      // two register sets: the loads have two rounds to arrive
            double *Aw = As[par ^ 1], *Bw = Bs[par ^ 1];
            if constexpr (par == 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { Aw[(sk0 + 2 * u) * LD + sr] = pa2[u]; Bw[(sk0 + 2 * u) * LD + sr] = pb2[u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { pa2[u] = ga[(size_t)(sk0 + 2 * u) * gstride]; pb2[u] = gb[(size_t)(sk0 + 2 * u) * gstride]; }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) { Aw[(sk0 + 2 * u) * LD + sr] = pa[u]; Bw[(sk0 + 2 * u) * LD + sr] = pb[u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { pa[u] = ga[(size_t)(sk0 + 2 * u) * gstride]; pb[u] = gb[(size_t)(sk0 + 2 * u) * gstride]; }
            }
            ga += 16 * gstride; gb += 16 * gstride;
        }
#pragma unroll
        for (int k4 = 0; k4 < KT; k4 += 4) {
            if constexpr (MODE == 0) {                 // MFMA only
                mm(av, bv);
            } else if constexpr (MODE == 1) {          // + reads that feed nothing
#pragma unroll
                for (int a = 0; a < 4; ++a) { an[a] = Bt[k4 * LD + a * 16]; bn[a] = At[k4 * LD + a * 16]; }
                mm(av, bv);
#pragma unroll
                for (int a = 0; a < 4; ++a) asm volatile("" ::"v"(an[a]), "v"(bn[a]));
            } else if constexpr (MODE == 2) {          // read -> wait -> MFMA (no pipeline)
#pragma unroll
                for (int a = 0; a < 4; ++a) { av[a] = Bt[k4 * LD + a * 16]; bv[a] = At[k4 * LD + a * 16]; }
                __builtin_amdgcn_sched_barrier(0);
                mm(av, bv);
                __builtin_amdgcn_sched_barrier(0);
            } else {                                    // one-step software pipeline (modes 3..6)
#pragma unroll
                for (int a = 0; a < 4; ++a) { an[a] = Bt[((k4 + 4) & 15) * LD + a * 16]; bn[a] = At[((k4 + 4) & 15) * LD + a * 16]; }
                __builtin_amdgcn_sched_barrier(0);
                mm(av, bv);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 4; ++a) { av[a] = an[a]; bv[a] = bn[a]; }
            }
        }
        if constexpr (MODE >= 4) __syncthreads();
    };
    for (int rd = 0; rd < rounds; rd += 2) { round(std::integral_constant<int, 0>{}, rd); round(std::integral_constant<int, 1>{}, rd + 1); }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    double s = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    out[blockIdx.x * 256 + tid] = s;
}

static double *g_src = nullptr; static long g_stride = 36608;
template <int MODE>
void run(int wg_per_cu, int rounds, const char *name) {
    const int grid = 256 * wg_per_cu;
    double *out;
    hipMalloc(&out, sizeof(double) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe<MODE>), dim3(grid), dim3(256), 0, 0, out, rounds, 1.0, g_src, g_stride);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double flops = 2.0 * 16 * 16 * 4 * 16.0 * 4 * rounds * 4.0 * grid;
    printf("%-34s WG/CU=%d : %8.3f ms %7.2f TFLOP/s\n", name, wg_per_cu, best, flops / best / 1e9);
    hipFree(out);
}

int main() {
    // global source for mode 6: 128 rounds x 16 columns of stride g_stride doubles (a panel of 4576 rows), 512 row blocks
    { const size_t n = (size_t)g_stride * 16 * 130 + 512 * 128 + 64 * 128 + 1024; hipMalloc(&g_src, n * 8); hipMemset(g_src, 0, n * 8); }
    for (int w = 1; w <= 2; ++w) {
        run<0>(w, 2000, "mfma only");
        run<1>(w, 2000, "mfma + idle LDS reads");
        run<2>(w, 2000, "LDS read -> wait -> mfma");
        run<3>(w, 2000, "LDS read one step ahead");
        run<4>(w, 2000, "  + barrier per round");
        run<5>(w, 2000, "  + 16 LDS writes per round");
        run<6>(w, 128, "  + 16 global loads per round");
        run<7>(w, 128, "  + loads two rounds ahead");
    }
    return 0;
}
