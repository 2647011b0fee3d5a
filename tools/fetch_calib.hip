// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// widths our kernels use (8 B/lane and 16 B/lane coalesced streams), as
// /opt/skills/guides/MI355X_MICROARCH.md (section HBM) asks before trusting an absolute.
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/fetch_calib     (and again with WRITE_SIZE)
// Each kernel streams exactly 4 GiB in (reduction) or 4 GiB in + 4 GiB out (copy).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read8(const double *p, double *out, size_t n) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 1.2345e300) out[0] = s;
}
__global__ void k_read16(const double2 *p, double *out, size_t n) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;
}
__global__ void k_copy8(const double *p, double *q, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) q[i] = p[i];
}
int main() {
    const size_t n = (size_t)1 << 29;       // 512 Mi doubles = 4 GiB
    double *a, *b;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
    hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const double2 *)a, b, n / 2);
    hipLaunchKernelGGL(k_copy8, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipDeviceSynchronize();
    printf("each kernel: 4 GiB read (k_copy8: + 4 GiB written)\n");
    return 0;
}
