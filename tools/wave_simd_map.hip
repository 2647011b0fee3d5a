// Which SIMD does wave w of a 512-thread workgroup run on?  (k_update's balanced block mapping for diagonal tiles assumes that waves
// w and w + 4 share SIMD w % 4.)  Prints count[wave][SIMD_ID] over many workgroups of the k_update shape (72 KB LDS, 8 waves).
//   hipcc --offload-arch=gfx950 -O2 -o tools/wave_simd_map tools/wave_simd_map.hip && tools/wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512, 4) void k(unsigned *out) {
    __shared__ double pad[9216];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID, all 32 bits
        out[(size_t)blockIdx.x * 8 + (threadIdx.x >> 6)] = hw + (pad[threadIdx.x] < 0 ? 1u : 0u);
    }
}
int main() {
    const int G = 4096;
    unsigned *d; hipMalloc(&d, (size_t)G * 8 * 4);
    hipLaunchKernelGGL(k, dim3(G), dim3(512), 0, 0, d);
    std::vector<unsigned> h((size_t)G * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    long cnt[8][4] = {};
    long same_cu = 0;
    for (int b = 0; b < G; ++b) {
        for (int w = 0; w < 8; ++w) cnt[w][(h[(size_t)b * 8 + w] >> 4) & 3]++;
        bool s = true;
        for (int w = 1; w < 8; ++w) s = s && (((h[(size_t)b * 8 + w] >> 8) & 0xff) == ((h[(size_t)b * 8] >> 8) & 0xff));
        same_cu += s;
    }
    printf("wave : SIMD0 SIMD1 SIMD2 SIMD3   (HW_ID bits 5:4) over %d workgroups of 8 waves\n", G);
    for (int w = 0; w < 8; ++w) printf("  %d  : %5ld %5ld %5ld %5ld\n", w, cnt[w][0], cnt[w][1], cnt[w][2], cnt[w][3]);
    printf("first workgroups, SIMD of waves 0..7:");
    for (int b = 0; b < 6; ++b) { printf("  ["); for (int w = 0; w < 8; ++w) printf("%u", (h[(size_t)b * 8 + w] >> 4) & 3); printf("]"); }
    printf("\n");
    return 0;
}
