// update_bench.hip -- isolates k_update (the fp64-MFMA panel update) on the shape that dominates
// config C4: 64 fronts of order f = 4525 with ns = 3530 pivots, trailing update after the first
// 256-column outer panel.  Build variants with -DUPD_VARIANT=n (see kernels.hip) to ablate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tulip.jl_amd/csrc tools/update_bench.hip -o tools/update_bench
#include "../tulip.jl_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
using namespace tlpk;

int main(int argc, char **argv) {
    const int nfronts = argc > 1 ? atoi(argv[1]) : 64;
    const i32 f = argc > 2 ? atoi(argv[2]) : 4525, ns = argc > 3 ? atoi(argv[3]) : 3530;
    // mode "right": trailing update after the first kw columns (K = kw, all trailing columns);
    // mode "left" (5th arg = 1): left-looking update of the block column [kw, kw+256) with K = [0, kw)
    const i32 k0 = 0, kw = argc > 4 ? atoi(argv[4]) : 256;
    const bool left = argc > 5 && atoi(argv[5]) == 1;
    const i32 rs = f - ns;
    std::vector<FrontDesc> fr(nfronts);
    i64 loff = 0, uoff = 0;
    for (int s = 0; s < nfronts; ++s) {
        fr[s] = FrontDesc{};
        fr[s].loff = loff; fr[s].uoff = uoff; fr[s].f = f; fr[s].ns = ns; fr[s].ubuf = 0; fr[s].parent = -1;
        fr[s].lda = (f + 15) / 16 * 16;
        loff += (i64)fr[s].lda * ns; uoff += (i64)rs * rs;
    }
    std::vector<UpdateTask> tasks;
    double flops = 0;
    for (int s = 0; s < nfronts; ++s) {
        const i32 c0 = k0 + kw, c1 = left ? std::min(c0 + 256, ns) : f;
        for (i32 cc = c0; cc < c1; ++cc) flops += 2.0 * kw * (double)(f - cc);
        for (i32 j0 = c0; j0 < c1; j0 += TILE)
            for (i32 i0 = j0; i0 < f; i0 += TILE) tasks.push_back(UpdateTask{s, k0, kw, i0, j0, c1, 0, 0});
    }
    DevCtx c{};
    FrontDesc *dfr; UpdateTask *dt; double *L, *U; int *info;
    hipMalloc(&dfr, sizeof(FrontDesc) * nfronts); hipMemcpy(dfr, fr.data(), sizeof(FrontDesc) * nfronts, hipMemcpyHostToDevice);
    hipMalloc(&dt, sizeof(UpdateTask) * tasks.size()); hipMemcpy(dt, tasks.data(), sizeof(UpdateTask) * tasks.size(), hipMemcpyHostToDevice);
    hipMalloc(&L, sizeof(double) * loff); hipMalloc(&U, sizeof(double) * (uoff + 1)); hipMalloc(&info, 16);
    {   // random-ish fill (not zeros: DVFS) via a tiny kernel-free host pattern
        std::vector<double> h((size_t)1 << 22);
        for (size_t i = 0; i < h.size(); ++i) h[i] = ((double)((i * 2654435761u) % 2001) - 1000.0) * 1e-3;
        for (i64 off = 0; off < loff; off += (i64)h.size())
            hipMemcpy(L + off, h.data(), sizeof(double) * std::min<i64>((i64)h.size(), loff - off), hipMemcpyHostToDevice);
        hipMemset(U, 0, sizeof(double) * (uoff + 1));
    }
    c.upd_remap = getenv("TLPK_UPD_REMAP") ? atoi(getenv("TLPK_UPD_REMAP")) : 0;
    { double *sp; hipMalloc(&sp, tasks.size() * 64 + 64); hipMemset(sp, 0, tasks.size() * 64 + 64); c.spart = sp; }
    c.fronts = dfr; c.Lval = L; c.U0 = U; c.U1 = U; c.info = info;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // optional 6th arg: number of streams the fronts are split over (concurrent-kernel test)
    const int nstreams = argc > 6 ? atoi(argv[6]) : 1;
    if (nstreams > 1) {
        std::vector<hipStream_t> st(nstreams);
        for (auto &x : st) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
        std::vector<hipEvent_t> done(nstreams);
        for (auto &x : done) hipEventCreate(&x);
        // tasks are front-major: split into contiguous chunks of fronts
        std::vector<size_t> cut(nstreams + 1, 0);
        for (int g = 1; g <= nstreams; ++g) {
            const int fcut = (int)((long long)nfronts * g / nstreams);
            size_t i = cut[g - 1];
            while (i < tasks.size() && tasks[i].front < fcut) ++i;
            cut[g] = i;
        }
        float bestm = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, st[0]);
            for (int g = 1; g < nstreams; ++g) hipStreamWaitEvent(st[g], e0, 0);
            for (int g = 0; g < nstreams; ++g) {
                hipLaunchKernelGGL(k_update<false>, dim3((unsigned)(cut[g + 1] - cut[g])), dim3(256), 0, st[g], dt + cut[g], c);
                hipEventRecord(done[g], st[g]);
            }
            for (int g = 1; g < nstreams; ++g) hipStreamWaitEvent(st[0], done[g], 0);
            hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < bestm) bestm = ms;
        }
        printf("  %d streams: %.3f ms  %.2f TFLOP/s\n", nstreams, bestm, flops / bestm / 1e9);
    }
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_update<false>, dim3((unsigned)tasks.size()), dim3(256), 0, 0, dt, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
#ifdef UPD_TRACE
    {
        std::vector<unsigned long long> tr(tasks.size() * 8);
        hipMemcpy(tr.data(), c.spart, tr.size() * 8, hipMemcpyDeviceToHost);
        char nm[256]; snprintf(nm, sizeof nm, "gpurun_out/upd_trace_%d_%d.csv", kw, (int)left);
        FILE *fp = fopen(nm, "w");
        if (fp) {
            fprintf(fp, "wg,t0,t1,t2,t3,hwid,xcc\n");
            for (size_t i = 0; i < tasks.size(); ++i)
                fprintf(fp, "%zu,%llu,%llu,%llu,%llu,%llu,%llu\n", i, tr[i*8], tr[i*8+1], tr[i*8+2], tr[i*8+3], tr[i*8+4], tr[i*8+5]);
            fclose(fp);
        }
    }
#endif
    printf("variant %d%s: fronts=%d f=%d ns=%d kw=%d tiles=%zu : %.3f ms  %.2f TFLOP/s (algorithmic)  %.2f TFLOP/s (executed)  %.1f us/tile-slot\n",
#ifdef UPD_VARIANT
           UPD_VARIANT,
#else
           0,
#endif
           left ? " (left-looking)" : "", nfronts, f, ns, kw, tasks.size(), best, flops / best / 1e9, (double)tasks.size() * 2.0 * TILE * TILE * kw / best / 1e9, best * 1e3 * 512.0 / (double)tasks.size());
    return 0;
}
