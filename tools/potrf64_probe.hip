// Probe: 64 x 64 Cholesky + explicit inverse by ONE wave without workgroup barriers -- lane r owns row r, 16-column panels in
// registers, broadcasts by v_readlane (no LDS round trip per column), left-looking panel updates and the off-diagonal blocks of
// the inverse on the matrix cores with operands in LDS.  Compared with the per-column LDS hand-over of potrf_block (kernels.hip):
// ~1050 cycles per column there (tools/potrf_chain_probe.hip), 64 barrier round trips per block.
//   hipcc --offload-arch=gfx950 -O3 -o tools/potrf64_probe tools/potrf64_probe.hip && tools/potrf64_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int NB = 64, PW = 16, LDL = NB + 16, LDT = 17;

__device__ __forceinline__ double rdlane(double v, int lane) {
    const long long x = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffll), lane), hi = __builtin_amdgcn_readlane((int)(x >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
#define LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__global__ __launch_bounds__(64) void potrf64(const double *__restrict__ Ain, double *__restrict__ Lout, double *__restrict__ Wout, long long *cyc, int reps) {
    __shared__ double Lt[NB * LDL];          // Lt[k * LDL + row] = L[row][k]
    __shared__ double Wr[NB * LDL];          // Wr[t * LDL + c] = W[t][c]
    __shared__ double Wd[4][PW * LDT];       // Wd[p][t * LDT + r] = W_pp[r][t]   (diagonal blocks, column-major)
    __shared__ double Ts[NB * LDT];          // transposition scratch: Ts[row * LDT + c]
    const int r = threadIdx.x, lr = r & 15, lk = r >> 4;
    long long t0 = 0, tph[6] = {0, 0, 0, 0, 0, 0}, tl = 0;
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long tn = clock64(); __builtin_amdgcn_sched_barrier(0); if (rep > 0) tph[i] += tn - tl; tl = tn; } while (0)
    for (int rep = 0; rep < reps; ++rep) {
    if (rep == 1) t0 = clock64();
    tl = clock64();
    for (int i = r; i < NB * LDL; i += 64) { Lt[i] = 0.0; Wr[i] = 0.0; }
    LDS_FENCE();
    STAMP(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        double a[PW], w[PW];
#pragma unroll
        for (int c = 0; c < PW; ++c) { a[c] = (r >= PW * p + c) ? Ain[r + (PW * p + c) * NB] : 0.0; w[c] = (r == PW * p + c) ? 1.0 : 0.0; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(1);
        if (p > 0) {
            // U[row][c] = sum_{k < 16p} L[row][k] L[16p + c][k] on the matrix cores, row blocks b >= p
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b < p) continue;
                v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k4 = 0; k4 < PW * p; k4 += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[(k4 + lk) * LDL + PW * p + lr], Lt[(k4 + lk) * LDL + PW * b + lr], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) Ts[(PW * b + lr) * LDT + lk + 4 * q] = acc[q];
            }
            LDS_FENCE();
#pragma unroll
            for (int c = 0; c < PW; ++c) a[c] -= (r >= PW * p) ? Ts[r * LDT + c] : 0.0;
        }
        STAMP(2);
        // panel: 16 column steps, broadcasts by readlane; [A_pp | I] elimination gives the inverse of the diagonal block
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int J = PW * p + j;
            double d = rdlane(a[j], J);
            if (!(d > 0.0)) d = 1.0;
            double isq = __builtin_amdgcn_rsq(d);
            isq = isq * (1.5 - 0.5 * d * isq * isq);
            isq = isq * (1.5 - 0.5 * d * isq * isq);
            double sq = d * isq;
            sq = fma(0.5 * isq, fma(-sq, sq, d), sq);
            const double inv2 = isq * isq;
            const double arj = (r > J) ? a[j] * inv2 : 0.0;
#pragma unroll
            for (int c = j + 1; c < PW; ++c) a[c] = fma(-arj, rdlane(a[j], PW * p + c), a[c]);
#pragma unroll
            for (int c = 0; c <= j; ++c) w[c] = fma(-arj, rdlane(w[c], J), w[c]);
            a[j] = (r == J) ? sq : ((r > J) ? a[j] * isq : a[j]);
        }
        STAMP(3);
        // diagonal of L for the row scaling of the inverse block
        double lii = 1.0;
#pragma unroll
        for (int c = 0; c < PW; ++c) lii = (r == PW * p + c) ? a[c] : lii;
        const double ili = 1.0 / lii;
        const bool inblk = (r >= PW * p) && (r < PW * p + PW);
#pragma unroll
        for (int c = 0; c < PW; ++c) {
            const int col = PW * p + c;
            const double lv = (r >= col) ? a[c] : 0.0;
            Lt[col * LDL + r] = lv;
            Lout[r + col * NB] = lv;
            if (inblk) {
                const double wv = (r >= col) ? w[c] * ili : 0.0;
                Wr[r * LDL + col] = wv;                        // row-major
                Wd[p][c * LDT + (r - PW * p)] = wv;            // column-major copy of the diagonal block
                Wout[r + col * NB] = wv;
            }
        }
        LDS_FENCE();
        STAMP(4);
    }
    // off-diagonal blocks of the inverse, by block distance: W_ij = -W_ii (sum_{k=j}^{i-1} L_ik W_kj)
#pragma unroll
    for (int dist = 1; dist < 4; ++dist) {
#pragma unroll
        for (int i = dist; i < 4; ++i) {
            const int j = i - dist;
            // G[r][c] = sum_k sum_t L_ik[r][t] W_kj[t][c]:  first operand M1[c][t] = W_kj[t][c], second M2[r][t] = L_ik[r][t]
            v4f64 g = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < j || k >= i) continue;
#pragma unroll
                for (int k4 = 0; k4 < PW; k4 += 4)
                    g = __builtin_amdgcn_mfma_f64_16x16x4f64(Wr[(PW * k + k4 + lk) * LDL + PW * j + lr], Lt[(PW * k + k4 + lk) * LDL + PW * i + lr], g, 0, 0, 0);
            }
            // g reg q of lane l: G[r = lr][c = lk + 4q]  ->  Ts[t = r][c]
#pragma unroll
            for (int q = 0; q < 4; ++q) Ts[lr * LDT + lk + 4 * q] = g[q];
            LDS_FENCE();
            // H[r][c] = sum_t W_ii[r][t] G[t][c]: first operand M1[c][t] = G[t][c], second M2[r][t] = W_ii[r][t]
            v4f64 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k4 = 0; k4 < PW; k4 += 4)
                h = __builtin_amdgcn_mfma_f64_16x16x4f64(Ts[(k4 + lk) * LDT + lr], Wd[i][(k4 + lk) * LDT + lr], h, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = PW * i + lr, cc = PW * j + lk + 4 * q;
                Wr[rr * LDL + cc] = -h[q];
                Wout[rr + cc * NB] = -h[q];
            }
            LDS_FENCE();
        }
    }
    STAMP(5);
    }
    if (r == 0) { cyc[0] = (clock64() - t0) / (reps > 1 ? reps - 1 : 1); for (int i = 0; i < 6; ++i) cyc[1 + i] = tph[i] / (reps > 1 ? reps - 1 : 1); }
}

int main() {
    std::vector<double> A(NB * NB), L(NB * NB, 0.0), W(NB * NB, 0.0), Lh(NB * NB, 0.0);
    // SPD test matrix: B B' + 64 I with an asymmetric B
    std::vector<double> B(NB * NB);
    for (int i = 0; i < NB; ++i) for (int j = 0; j < NB; ++j) B[i + j * NB] = std::sin(0.37 * i + 1.3 * j) + 0.01 * i - 0.02 * j;
    for (int i = 0; i < NB; ++i) for (int j = 0; j < NB; ++j) { double s = 0; for (int k = 0; k < NB; ++k) s += B[i + k * NB] * B[j + k * NB]; A[i + j * NB] = s + (i == j ? 64.0 : 0.0); }
    // host Cholesky
    Lh = A;
    for (int j = 0; j < NB; ++j) {
        double d = Lh[j + j * NB]; for (int k = 0; k < j; ++k) d -= Lh[j + k * NB] * Lh[j + k * NB];
        d = std::sqrt(d); Lh[j + j * NB] = d;
        for (int i = j + 1; i < NB; ++i) { double s = Lh[i + j * NB]; for (int k = 0; k < j; ++k) s -= Lh[i + k * NB] * Lh[j + k * NB]; Lh[i + j * NB] = s / d; }
    }
    double *dA, *dL, *dW; long long *dc;
    hipMalloc(&dA, NB * NB * 8); hipMalloc(&dL, NB * NB * 8); hipMalloc(&dW, NB * NB * 8); hipMalloc(&dc, 64);
    hipMemcpy(dA, A.data(), NB * NB * 8, hipMemcpyHostToDevice);
    hipMemset(dL, 0, NB * NB * 8); hipMemset(dW, 0, NB * NB * 8);
    hipLaunchKernelGGL(potrf64, dim3(1), dim3(64), 0, 0, dA, dL, dW, dc, 9);
    long long c = 0, ph[8];
    hipMemcpy(L.data(), dL, NB * NB * 8, hipMemcpyDeviceToHost); hipMemcpy(W.data(), dW, NB * NB * 8, hipMemcpyDeviceToHost); hipMemcpy(ph, dc, 56, hipMemcpyDeviceToHost); c = ph[0];
    printf("cycles: zero-fill %lld, panel loads %lld, matrix-core panel updates + transposition %lld, 64 column steps %lld, stores %lld, off-diagonal inverse blocks %lld\n", ph[1], ph[2], ph[3], ph[4], ph[5], ph[6]);
    double eL = 0, eW = 0;
    for (int j = 0; j < NB; ++j) for (int i = j; i < NB; ++i) eL = std::fmax(eL, std::fabs(L[i + j * NB] - Lh[i + j * NB]));
    // W L = I ?
    for (int i = 0; i < NB; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = j; k <= i; ++k) s += W[i + k * NB] * Lh[k + j * NB]; eW = std::fmax(eW, std::fabs(s - (i == j ? 1.0 : 0.0))); }
    printf("64 x 64 potrf + inverse in one wave: %lld cycles (%.1f us at 2.33 GHz); max |L - L_host| = %.2e, max |W L - I| = %.2e\n", c, c / 2330.0, eL, eW);
    return 0;
}
