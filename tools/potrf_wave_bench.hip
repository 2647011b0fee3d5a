// potrf_wave_bench.hip -- the 64 x 64 diagonal-block factorisations of kernels.hip in isolation: NF independent dense fronts of order 64 (k_potrf) or 256
// (k_potrf_wide), potrf_block (256 threads, one barrier per column) against potrf_block_wave (one wave, readlane broadcasts) and potrf_block_pair (round 5: one barrier per two columns, bit-identical to potrf_block) and potrf_block_dpp (round 5: DPP broadcasts); -DPOTRF_TRACE adds
// 100 MHz phase stamps of the wave version.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPOTRF_TRACE -I tulip.jl_amd/csrc tools/potrf_wave_bench.hip -o tools/potrf_wave_bench
#include "../tulip.jl_amd/csrc/kernels.hip"
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
using namespace tlpk;

template <int MODE>
static double run(int nf, int n, const std::vector<double> &A0, std::vector<double> &Lout, std::vector<unsigned long long> *stamps, std::vector<double> *Wout = nullptr) {
    std::vector<FrontDesc> fr(nf);
    i64 loff = 0, doff = 0;
    const i32 lda = (n + 15) / 16 * 16;
    for (int s = 0; s < nf; ++s) {
        fr[s] = FrontDesc{}; fr[s].loff = loff; fr[s].f = n; fr[s].ns = n; fr[s].lda = lda; fr[s].parent = -1; fr[s].dinvoff = doff;
        loff += pk_len(lda, n); doff += (i64)((n + 63) / 64) * 64 * 64;
    }
    std::vector<double> host((size_t)loff, 0.0);
    for (int s = 0; s < nf; ++s)
        for (int c = 0; c < n; ++c) for (int r = c; r < n; ++r) host[(size_t)(fr[s].loff + pk_off(lda, c) + r)] = A0[(size_t)r + (size_t)c * n];
    std::vector<PotrfTask> tasks(nf);
    for (int s = 0; s < nf; ++s) tasks[s] = PotrfTask{s, 0, n, 0};
    DevCtx c{};
    FrontDesc *dfr; PotrfTask *dt; double *L, *dinv, *sp; int *info;
    hipMalloc(&dfr, sizeof(FrontDesc) * nf); hipMemcpy(dfr, fr.data(), sizeof(FrontDesc) * nf, hipMemcpyHostToDevice);
    hipMalloc(&dt, sizeof(PotrfTask) * nf); hipMemcpy(dt, tasks.data(), sizeof(PotrfTask) * nf, hipMemcpyHostToDevice);
    hipMalloc(&L, 8 * loff); hipMalloc(&dinv, 8 * doff); hipMemset(dinv, 0, 8 * doff); hipMalloc(&sp, 8 * 32 * nf + 64); hipMalloc(&info, 16);
    hipMemset(sp, 0, 8 * 32 * nf + 64);
    { int big = 0x7fffffff; hipMemcpy(info, &big, 4, hipMemcpyHostToDevice); }
    c.fronts = dfr; c.Lval = L; c.dinv = dinv; c.info = info; c.spart = sp;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemcpy(L, host.data(), 8 * loff, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (n <= 64) hipLaunchKernelGGL((k_potrf<false, MODE>), dim3(nf), dim3(256), 0, 0, dt, c);
        else hipLaunchKernelGGL((k_potrf_wide<false, MODE>), dim3(nf), dim3(256), 0, 0, dt, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    Lout.resize((size_t)loff); hipMemcpy(Lout.data(), L, 8 * loff, hipMemcpyDeviceToHost);
    if (Wout) { Wout->resize((size_t)doff); hipMemcpy(Wout->data(), dinv, 8 * doff, hipMemcpyDeviceToHost); }
    if (stamps) { stamps->resize(32); hipMemcpy(stamps->data(), sp, 8 * 32, hipMemcpyDeviceToHost); }
    hipFree(dfr); hipFree(dt); hipFree(L); hipFree(dinv); hipFree(sp); hipFree(info);
    return best * 1e3;
}

int main(int argc, char **argv) {
    const int nf = argc > 1 ? atoi(argv[1]) : 64, n = argc > 2 ? atoi(argv[2]) : 64;
    std::vector<double> B((size_t)n * n), A((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) B[i + (size_t)j * n] = std::sin(0.37 * i + 1.3 * j) + 0.01 * i - 0.02 * j;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += B[i + (size_t)k * n] * B[j + (size_t)k * n]; A[i + (size_t)j * n] = s + (i == j ? n : 0.0); }
    std::vector<double> L0, L1, L2, L3, W0, W3; std::vector<unsigned long long> st;
    const double t0 = run<0>(nf, n, A, L0, nullptr, &W0), t1 = run<1>(nf, n, A, L1, &st), t2 = run<2>(nf, n, A, L2, nullptr), t3 = run<3>(nf, n, A, L3, &st, &W3);
    double err = 0, mx = 0, err3 = 0, werr = 0, wmx = 0; size_t ndiff = 0, nan3 = 0;
    for (size_t i = 0; i < L0.size(); ++i) {
        err = std::fmax(err, std::fabs(L0[i] - L1[i])); mx = std::fmax(mx, std::fabs(L0[i])); ndiff += (std::memcmp(&L0[i], &L2[i], 8) != 0);
        if (std::isnan(L3[i])) ++nan3; else err3 = std::fmax(err3, std::fabs(L0[i] - L3[i]));
    }
    for (size_t i = 0; i < W0.size(); ++i) { if (std::isnan(W3[i])) ++nan3; else werr = std::fmax(werr, std::fabs(W0[i] - W3[i])); wmx = std::fmax(wmx, std::fabs(W0[i])); }
    printf("%d fronts of order %d: potrf_block %.1f us, potrf_block_wave %.1f us, potrf_block_pair %.1f us, potrf_block_dpp %.1f us per launch; max |L0 - L_wave| = %.2e, max |L0 - L_dpp| = %.2e (max |L| %.2e), "
           "max |W0 - W_dpp| = %.2e (max |W| %.2e), NaNs in the dpp results: %zu; entries of L_pair that differ from L0 in any bit: %zu of %zu\n", nf, n, t0, t1, t2, t3, err, err3, mx, werr, wmx, nan3, ndiff, L0.size());
    printf("dpp stamps of workgroup 0 (10 ns units, differences):");
    for (int i = 1; i < 32 && st[i]; ++i) printf(" %llu", st[i] - st[i - 1]);
    printf("\n");
    if (st.size() >= 24 && st[16] && st[23]) {            // k_potrf_wide, mode 3: potrf | solve | potrf | solve | potrf | solve | potrf (phase boundaries of the block column)
        printf("block-column phases of workgroup 0 behind its first block (10 ns units: barrier, solve 0, block 1, solve 1, block 2, solve 2, block 3):");
        for (int i = 17; i < 24; ++i) printf(" %llu", st[i] - st[i - 1]);
        printf("\n");
    }
    return 0;
}
