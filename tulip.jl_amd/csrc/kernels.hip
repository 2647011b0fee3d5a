// kernels.hip -- gfx950 (CDNA4, wave64) device kernels of libtlpk: numeric phase of the
// normal-equations Newton step.
//
//   update!  (/root/reference/src/KKT/Cholmod/spd.jl:22-50)
//     k_compute_d     D = 1/(theta_inv + regP)                                  spd.jl:42
//     k_assemble      S = A*D*A' + diag(regD) gathered straight into the supernodal panels  spd.jl:43
//     k_single_factor isolated 1 x 1 fronts, one thread each
//     k_extend_add    multifrontal assembly of the children's update matrices
//     k_potrf / k_potrf_wide / k_potrf_small, k_trsm / k_trsm_thin, k_update (+ k_update_reduce)
//                     blocked dense partial Cholesky of every front, per 256-wide block column:
//                     left-looking rank-K update on v_mfma_f64_16x16x4_f64, diagonal block in one
//                     workgroup (64-wide register-resident steps + inverses), rows below in one
//                     register-resident MFMA pass                                spd.jl:46
//   solve!   (spd.jl:52-70)
//     k_rhs           xi = xi_p + A*(D.*xi_d), permuted                          spd.jl:56-57
//     k_single_solve, k_fwd_gather, k_fwd_small / k_bwd_small, k_fwd_sweep / k_bwd_sweep (persistent: one launch per
//                     tree level and direction; k_fwd_diag / k_fwd_update / k_bwd_update: one launch per block step,
//                     TLPK_SWEEP=0)   supernodal forward / backward substitution   spd.jl:61
//   K2 (sqd.jl:24-74): the factor kernels instantiated SIGNED (P K P' = L S L'), k_k2_diag, k_k2_rhs, k_apply_signs, k_k2_out
//     k_unpermute, k_dx  dy = P' x ;  dx = D.*(A'dy - xi_d)                      spd.jl:64-66
//
// Two rules learnt by measurement run through this file: (1) never guard a load that feeds the next
// instruction (`c ? M[i] : 0`): clamp the address and select afterwards, or every load waits for the
// previous one; (2) ordinary stores of one workgroup are NOT visible to another workgroup of the same launch
// (the L2s of the 8 dies are not coherent for them, a CU's L1 never is): whatever crosses workgroups either
// crosses a launch, or -- the solve sweeps -- goes through relaxed agent-scope atomic stores and loads on both
// sides with the data as its own flag (cdna_hip_programming.md, Guideline 16).
//
// Every kernel is deterministic: sums that cross workgroups are ordered by the static schedule built on the host
// (symbolic.cpp: build_schedule).  The one floating-point atomic in this file -- k_update's epilogue, a fire-and-forget fp64 add executed
// by the L2 -- is deterministic because the schedule gives every target entry exactly ONE adder per launch (one tile per target block and
// launch; split-K parts go to scratch and are summed in order by k_update_reduce): the add is then the same IEEE operation as load /
// subtract / store, without the wave waiting for the old value.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "tlpk_device.hpp"

namespace tlpk {


typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_f64(const double v, const int l) {      // l wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double *front_u(const DevCtx &c, const FrontDesc &fd) {
    return (fd.ubuf ? c.U1 : c.U0) + fd.uoff;
}
// Packed panels (tlpk_host.hpp: pk_off): a panel is stored by 64-column slices, slice b from its first row 64 b down with leading
// dimension lda - 64 b.  pcol = pointer to the VIRTUAL row 0 of a panel column (valid for rows >= 64 (col / 64)); consecutive
// columns of one slice are pld apart.
__device__ __forceinline__ double *pcol(const DevCtx &c, const FrontDesc &fd, i32 col) { return c.Lval + fd.loff + pk_off(fd.lda, col); }
__device__ __forceinline__ i32 pld(const FrontDesc &fd, i32 col) { return fd.lda - ((col >> 6) << 6); }

// ------------------------------------------------------------------------------------------
// elementwise / sparse kernels
// ------------------------------------------------------------------------------------------
__global__ void k_compute_d(i64 n, const double *__restrict__ theta, const double *__restrict__ regP,
                            double *__restrict__ D) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) D[j] = 1.0 / (theta[j] + regP[j]);
}

// One thread per stored entry of S (lower triangle): value = sum_t w_t * D[j_t] (+ regD[i] on the
// diagonal), written to its slot in the panel storage.  Gather formulation: no atomics, fixed
// summation order.
__device__ __forceinline__ double asm_value(const i64 e, const i32 *__restrict__ diag_row, const i64 *__restrict__ pptr, const double *__restrict__ pw,
                                            const i32 *__restrict__ pj, const double *__restrict__ D, const double *__restrict__ regD) {
    const i64 p0 = pptr[e], p1 = pptr[e + 1];
    double s = 0.0;
    // four products per trip, indices clamped and weights zeroed past the end: the index -> D[j]
    // chains of a trip are independent loads (one product per trip waited for each chain in turn);
    // same summation order
    for (i64 p = p0; p < p1; p += 4) {
        double w[4]; i32 j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const i64 q = min(p + u, p1 - 1); w[u] = pw[q]; j[u] = pj[q]; }
        double d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = D[j[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += ((p + u < p1) ? w[u] : 0.0) * d[u];
    }
    const i32 dr = diag_row[e];
    if (dr >= 0) s += regD[dr];
    return s;
}
// target < 0: the entry belongs to a panel that k_front_assemble forms
// upper != nullptr: only the entries with upper[e] == part (the fronts whose zero-fill + assembly run beside the leaf levels: symbolic.cpp step 13d)
__global__ void k_assemble(i64 nent, const i64 *__restrict__ target, const i32 *__restrict__ diag_row,
                           const i64 *__restrict__ pptr, const double *__restrict__ pw,
                           const i32 *__restrict__ pj, const double *__restrict__ D,
                           const double *__restrict__ regD, double *__restrict__ Lval, const unsigned char *__restrict__ upper, int part) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nent) return;
    if (upper && (int)upper[e] != part) return;
    const i64 tg = target[e];
    if (tg < 0) return;
    Lval[tg] = asm_value(e, diag_row, pptr, pw, pj, D, regD);
}

// Isolated 1 x 1 fronts, one thread each: L = sqrt(s), and the whole solve x = b / s in one step
// (nothing else reads or writes their entries).
// csign != nullptr (K2): the pivot must carry the sign of its node; L = sqrt|d|.  The solve below stays x = b / |d|,
// the sign is applied with everybody else's between the sweeps (k_apply_signs).
__global__ void k_single_factor(i64 n, const i64 *__restrict__ loff, const i64 *__restrict__ dinvoff,
                                const i32 *__restrict__ col, double *__restrict__ Lval, double *__restrict__ dinv, int *info,
                                const double *__restrict__ csign) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = Lval[loff[i]];
    const double sj = csign ? csign[col[i]] : 1.0;
    if (!(sj * d > 0.0)) { atomicMin(info, col[i]); d = 1.0; }      // same convention as potrf_block
    d = fabs(d);
    const double l = sqrt(d);
    Lval[loff[i]] = l;
    dinv[dinvoff[i]] = 1.0 / l;
}
__global__ void k_single_solve(i64 n, const i64 *__restrict__ dinvoff, const i32 *__restrict__ col,
                               const double *__restrict__ dinv, double *__restrict__ xw, i64 xw2) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (blockIdx.y) xw += xw2;                                    // (grid y = right-hand side of a pair)
    const double w = dinv[dinvoff[i]];
    xw[col[i]] = (xw[col[i]] * w) * w;                            // forward (x L^-1) then backward (x L^-1)
}

// Zero-fill of the factor storage before the assembly.  One workgroup per 64-column slice of a panel: rows from the slice's
// first row down to lda -- all a slice stores (packed panels, tlpk_host.hpp: the blocks ABOVE the 64 x 64 diagonal blocks, 39 % of
// a square pivot block, are read by no kernel -- updates, trsm, extend-add, assembly and the sweeps address rows >= the block's
// first row only -- and have no storage).
__global__ __launch_bounds__(256) void k_zero_panels(const i32 *__restrict__ tasks, DevCtx c) {
    const i32 s = tasks[2 * blockIdx.x], c0 = tasks[2 * blockIdx.x + 1];
    const FrontDesc fd = c.fronts[s];
    const i32 lda = fd.lda, nc = min(NB_IN, fd.ns - c0), ld = pld(fd, c0);          // c0 = first column of a slice
    double *P = pcol(c, fd, c0);
    // gridDim.y > 1: the slice's rows in gridDim.y pieces (multiples of 256 rows) -- short workgroups, so that the zero-fill of the upper fronts
    // (symbolic.cpp step 13d) leaves slots to the leaf levels' launches that run beside it
    const i32 rows = lda - c0, per = ((rows + (i32)gridDim.y - 1) / (i32)gridDim.y + 255) & ~255;
    const i32 rb = c0 + (i32)blockIdx.y * per, re = min(lda, rb + per);
    for (i32 col = 0; col < nc; ++col)
        for (i32 r = rb + (i32)threadIdx.x; r < re; r += 256) P[(i64)col * ld + r] = 0.0;
}

__global__ __launch_bounds__(256) void k_zero_small(const i32 *__restrict__ fronts, i64 n, DevCtx c) {      // one wave per small front
    const i64 idx = (i64)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (idx >= n) return;
    const FrontDesc fd = c.fronts[fronts[idx]];
    const i64 len = (i64)fd.lda * fd.ns;
    double *P = c.Lval + fd.loff;
    for (i64 e = threadIdx.x & 63; e < len; e += 64) P[e] = 0.0;
}

// ------------------------------------------------------------------------------------------
// extend-add: one workgroup owns parent columns [j0, j1) and adds, child after child, the child
// update-matrix columns that land in its range (panel columns before the front is factorised, U
// columns after its U has been written by k_update).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_extend_add(const EaTask *__restrict__ tasks, DevCtx c) {
    // Parent column tc of the range is owned by wave (tc - j0) & 3 for the whole kernel: every
    // contribution to a column is applied by the same wave in child order (deterministic) and the
    // four waves never need a barrier inside a batch of children.
    // The per-child lookups (descriptor + two binary searches in its relative-index list: ~25
    // dependent loads) are done for a whole batch of children at once, one child per thread;
    // doing them child after child made this kernel latency-bound (6 ms at 1.5 TB/s on config C4).
    constexpr int CB = 256;                         // children per batch
    __shared__ i32 s_q0[CB], s_q1[CB], s_rsc[CB], s_rlo[CB], s_rhi[CB];
    __shared__ i64 s_uoff[CB], s_reloff[CB];
    __shared__ int s_ubuf[CB];
    __shared__ unsigned char s_tc[CB][EA_COLS];     // parent column - j0 of the child's columns [q0, q1)
    const EaTask t = tasks[blockIdx.x];
    const FrontDesc fd = c.fronts[t.front];
    const i32 f = fd.f, ns = fd.ns, rs = f - ns;
    double *Up = front_u(c, fd);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (i32 cb = 0; cb < fd.nchild; cb += CB) {
        const i32 nb = min(CB, fd.nchild - cb);
        if (cb > 0) __syncthreads();                // previous batch fully consumed
        if (tid < nb) {
            const FrontDesc cd = c.fronts[c.children[fd.child_ptr + cb + tid]];
            const i32 rsc = cd.f - cd.ns;
            const i32 *relc = c.rel + cd.reloff;
            // child columns whose parent column lies in [j0, j1): two entries of the lookup table built by the analyse
            // phase (two binary searches in the child's list here cost ~20 dependent loads per workgroup), then the
            // <= EA_COLS parent columns themselves, loaded side by side
            const i32 q0 = c.ea_tab[cd.eatab + t.bidx], q1 = c.ea_tab[cd.eatab + t.bidx + 1];
            s_q0[tid] = q0; s_q1[tid] = q1; s_rsc[tid] = rsc;
            // row band of the task (boundaries br0 .. br1 of the parent's ranges; br1 == 0: every row): the child's rows inside it, from the same table
            s_rlo[tid] = t.br1 ? c.ea_tab[cd.eatab + t.br0] : 0; s_rhi[tid] = t.br1 ? c.ea_tab[cd.eatab + t.br1] : rsc;
            s_uoff[tid] = cd.uoff; s_reloff[tid] = cd.reloff; s_ubuf[tid] = cd.ubuf;
#pragma unroll
            for (int u = 0; u < EA_COLS; ++u) if (q0 + u < q1) s_tc[tid][u] = (unsigned char)(relc[q0 + u] - t.j0);
        }
        __syncthreads();
        for (i32 ci = 0; ci < nb; ++ci) {
            const i32 q0 = s_q0[ci], q1 = s_q1[ci];
            if (q0 >= q1) continue;
            const i32 rsc = s_rsc[ci], rlo = s_rlo[ci], rhi = s_rhi[ci];
            if (rlo >= rhi) continue;
            const double *Uc = (s_ubuf[ci] ? c.U1 : c.U0) + s_uoff[ci];
            const i32 *relc = c.rel + s_reloff[ci];
            for (i32 q = q0; q < q1; ++q) {
                const i32 tc = t.j0 + s_tc[ci][q - q0];
                if (((tc - t.j0) & 3) != wave) continue;
                const double *__restrict__ src = Uc + (i64)q * rsc;
                double *__restrict__ dst = (tc < ns) ? pcol(c, fd, tc) : (Up + (i64)(tc - ns) * rs - ns);
                // targets of one column are distinct rows: batches of 4 x 64 independent read-modify-writes, short
                // columns included (guards instead of a one-by-one tail: every trip is two dependent round trips).
                // Tried and measured slower: fire-and-forget L2 adds (8.9 vs 7.1 ms: the scattered targets cost the L2
                // more than the old-value round trip costs the waves); a flattened trip iterator with the loads of
                // trip n + 1 issued before the read-modify-write of trip n (6.3 vs 5.5 ms: the per-trip bookkeeping
                // costs more than the hidden round trip).
                for (i32 r = max(q, rlo) + lane; r < rhi; r += 256) {
                    i32 tg[4]; double v[4], d[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const i32 ru = min(r + 64 * u, rhi - 1);
                        tg[u] = relc[ru]; v[u] = src[ru];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) d[u] = dst[tg[u]];
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (r + 64 * u < rhi) dst[tg[u]] = d[u] + v[u];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Front assembly of the LARGE fronts (f >= fa_min_f(), symbolic.cpp): a workgroup FORMS one tile of the panel -- FA_CW = 16 parent
// columns x <= 256 rows -- in LDS: zero, the entries of S = A D A' + Rd that land in it (the arithmetic of k_assemble), then the
// children's update matrices in child order (the arithmetic and the order of k_extend_add), and writes the tile to the panel once.
// Against zero-fill + k_assemble + k_extend_add on the panel: no zero-fill, no read of the panel, and the ~12 contributions an entry of a
// big front receives from different children meet in LDS instead of pulling the entry's 128-byte line through the fabric 12 times
// (round 3: 26 GB moved for 7.5 GB of algorithmic extend-add bytes on config C4).  Deterministic: a parent column belongs to one
// wave for the whole tile, the targets of one (child, column) are distinct rows, LDS operations of a wave execute in order.
// The extend-add lookup table serves both directions: boundary k of the parent's ranges -> first child row at or after it.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ i32 fa_bound(const FrontDesc &p, const i32 k) {          // symbolic.cpp: ea_bound with FA_CW columns
    const i32 npan = (p.ns + FA_CW - 1) / FA_CW;
    return (k < npan) ? k * FA_CW : min(p.f, p.ns + (k - npan) * FA_CW);
}
constexpr int FA_RH = FA_RB * FA_CW;        // rows of a tile (at most)
__global__ __launch_bounds__(256) void k_front_assemble(const FaTask *__restrict__ tasks, DevCtx c, const i64 *__restrict__ colptr,
                                                        const i64 *__restrict__ target, const i32 *__restrict__ diag_row,
                                                        const i64 *__restrict__ pptr, const double *__restrict__ pw, const i32 *__restrict__ pj,
                                                        const double *__restrict__ D, const double *__restrict__ regD) {
    constexpr int CB = 128;                          // children per lookup batch
    __shared__ double tile[FA_CW][FA_RH];            // tile[parent column - j0][parent row - i0]
    __shared__ i32 s_q0[CB], s_q1[CB], s_r0[CB], s_r1[CB], s_rsc[CB];
    __shared__ i64 s_uoff[CB], s_reloff[CB];
    __shared__ int s_ubuf[CB];
    __shared__ unsigned char s_tc[CB][FA_CW];        // parent column - j0 of the child's columns [q0, q1)
    const FaTask t = tasks[blockIdx.x];
    const FrontDesc fd = c.fronts[t.front];
    const i32 ns = fd.ns;
    const i32 j0 = t.bc * FA_CW, j1 = min(j0 + FA_CW, ns);
    const i32 i0 = fa_bound(fd, t.br0), i1 = fa_bound(fd, t.br1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < FA_CW * FA_RH; e += 256) (&tile[0][0])[e] = 0.0;
    __syncthreads();
    // entries of S: column tc of the front = column col0 + tc of the permuted S, its entries are consecutive in the assembly list
    // with ascending rows; the position in the front is the target minus the address of the column's (virtual) row 0
    for (i32 tc = j0 + wave; tc < j1; tc += 4) {
        const i64 e0 = colptr[fd.col0 + tc], e1 = colptr[fd.col0 + tc + 1];
        const i64 base = fd.loff + pk_off(fd.lda, tc);
        for (i64 e = e0 + lane; e < e1; e += 64) {
            const i32 pos = (i32)(target[e] - base);
            if (pos >= i0 && pos < i1) tile[tc - j0][pos - i0] = asm_value(e, diag_row, pptr, pw, pj, D, regD);
        }
    }
    __syncthreads();
    for (i32 cb = 0; cb < fd.nchild; cb += CB) {
        const i32 nb = min(CB, fd.nchild - cb);
        if (cb > 0) __syncthreads();                // previous batch fully consumed
        if (tid < nb) {
            const FrontDesc cd = c.fronts[c.children[fd.child_ptr + cb + tid]];
            const i32 *relc = c.rel + cd.reloff;
            const i32 *tab = c.ea_tab + cd.eatab;
            const i32 q0 = tab[t.bc], q1 = tab[t.bc + 1];
            s_q0[tid] = q0; s_q1[tid] = q1; s_r0[tid] = tab[t.br0]; s_r1[tid] = tab[t.br1]; s_rsc[tid] = cd.f - cd.ns;
            s_uoff[tid] = cd.uoff; s_reloff[tid] = cd.reloff; s_ubuf[tid] = cd.ubuf;
#pragma unroll
            for (int u = 0; u < FA_CW; ++u) if (q0 + u < q1) s_tc[tid][u] = (unsigned char)(relc[q0 + u] - j0);
        }
        __syncthreads();
        for (i32 ci = 0; ci < nb; ++ci) {
            const i32 q0 = s_q0[ci], q1 = s_q1[ci], r0 = s_r0[ci], r1 = s_r1[ci];
            if (q0 >= q1 || r0 >= r1) continue;
            const i32 rsc = s_rsc[ci];
            const double *Uc = (s_ubuf[ci] ? c.U1 : c.U0) + s_uoff[ci];
            const i32 *relc = c.rel + s_reloff[ci];
            for (i32 q = q0; q < q1; ++q) {
                const i32 tcl = s_tc[ci][q - q0];
                if ((tcl & 3) != wave) continue;
                const double *__restrict__ src = Uc + (i64)q * rsc;
                double *__restrict__ dcol = tile[tcl] - i0;
                // rows of the child inside the tile, on or below the child's diagonal: batches of 4 x 64 independent loads, then
                // the adds (distinct rows of one LDS column)
                for (i32 r = max(r0, q) + lane; r < r1; r += 256) {
                    i32 tg[4]; double v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const i32 ru = min(r + 64 * u, r1 - 1);
                        tg[u] = relc[ru]; v[u] = src[ru];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (r + 64 * u < r1) dcol[tg[u]] += v[u];
                }
            }
        }
    }
    __syncthreads();
    // the tile, once: every stored entry of its columns in rows [i0, i1) -- the packed panel keeps a column from the first row of its
    // 64-column slice down (entries above the diagonal inside the 64 x 64 diagonal blocks: zero, as after the old zero-fill)
    const i32 rfirst = max(i0, (j0 >> 6) << 6);
    for (i32 tc = j0 + wave; tc < j1; tc += 4) {
        double *P = pcol(c, fd, tc);
        for (i32 r = rfirst + lane; r < i1; r += 64) P[r] = tile[tc - j0][r - i0];
    }
}

// ------------------------------------------------------------------------------------------
// potrf: Cholesky of one nb x nb diagonal block (nb <= NB_IN) by one workgroup, in LDS, followed
// by the inverse of the triangular block (used by the MFMA trsm and by the solve kernels).
// Right-looking; one barrier per column.  A pivot that is <= 0 or NaN records its (permuted)
// column in info[0] (min over all failures) and is replaced by 1 so that no NaN is produced;
// the host then reports TLPK_NOT_POSDEF (spd.jl:46-47).
// The inverse comes from applying the same row eliminations to an identity block (all 256
// threads, no extra barriers): that yields L~^{-1} of the unit-lower factor, then rows are scaled
// by 1/L_ii.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double *front_dinv(const DevCtx &c, const FrontDesc &fd, i32 k0) {
    return c.dinv + fd.dinvoff + (i64)(k0 / NB_IN) * (NB_IN * NB_IN);
}

constexpr int POTRF_SCRATCH = NB_IN * (NB_IN + 1) + 5 * NB_IN;   // doubles of LDS scratch

// SIGNED (augmented system K2): P K P' = L S L' with S = diag(c.csign) = +-1 known per column.  The unit-lower
// eliminations a_rc -= a_rj a_cj / d are sign-agnostic; a pivot must have the sign of its node (otherwise the
// matrix is not quasi-definite: reported like a non-positive pivot), the column is scaled by s_j / sqrt|d|, and
// products of two factor blocks carry S of the contracted columns (X S X').
template <bool SIGNED = false>
__device__ __forceinline__ void potrf_block(const DevCtx &c, const FrontDesc &fd, const i32 bk0, const i32 nb,
                                            const i32 kprev, double *scratch) {
    const double *sg = SIGNED ? c.csign + fd.col0 : nullptr;
    // Register-resident: thread (r, cg) owns A[r][col] and W[r][col] for col = cg + 4q, q = 0..15.
    // Per step only the pivot column of A and the pivot row of W go through LDS (double-buffered
    // by step parity => one barrier per column and no dependent LDS read-modify-write chains).
    double *Ds = scratch;                                   // NB_IN x (NB_IN + 1)
    double (*colbuf)[NB_IN] = reinterpret_cast<double (*)[NB_IN]>(scratch + NB_IN * (NB_IN + 1));
    double (*rowbuf)[NB_IN] = colbuf + 2;
    double *dg = scratch + NB_IN * (NB_IN + 1) + 4 * NB_IN;
    const i32 lda = pld(fd, bk0);                   // the block lies in ONE 64-column slice of the packed panel (bk0 is a multiple of 64): its leading dimension
    double *P = pcol(c, fd, bk0) + bk0;             // origin (bk0, bk0)
    const int tid = threadIdx.x, r = tid & 63, cg = tid >> 6;
    const bool rok = r < nb;
    double av[16], wv[16];
    // Left-looking inside the block column: subtract X X' where X = L[k0.., kprev..k0) are the
    // already factored 64-wide steps of this block column.  Wave w computes the 16 x 64 row strip
    // w of the 64 x 64 product on the matrix cores (operands straight from L2), the result goes
    // through LDS into the per-thread layout.
    const i32 Kp = bk0 - kprev;
    if (Kp > 0) {
        const int lane = tid & 63, lr = lane & 15, lk = lane >> 4;
        // X[rr][k] = panel(bk0 + rr, kprev + k): columns of earlier slices, 16 at a time inside one slice
        v4f64 dacc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) dacc[a] = (v4f64){0.0, 0.0, 0.0, 0.0};
        const i32 rrow = 16 * cg + lr;                                           // this wave's rows
        // rows are clamped instead of guarded (a clamped row only feeds entries that are never read),
        // and 4 k-steps of operands (20 loads) are issued before their MFMAs: the loop used to wait
        // for an L2 round trip per k-step, on the factorisation's critical path
        const i32 rr_c = min(rrow, nb - 1);
        i32 cr_c[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) cr_c[a] = min(16 * a + lr, nb - 1);
        for (i32 ks = 0; ks < Kp; ks += 16) {             // Kp is a multiple of NB_IN
            double bq[4], aq[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double *Xc = pcol(c, fd, kprev + ks + 4 * u + lk) + bk0;
                bq[u] = Xc[rr_c];
                if (SIGNED) bq[u] *= sg[kprev + ks + 4 * u + lk];
#pragma unroll
                for (int a = 0; a < 4; ++a) aq[u][a] = Xc[cr_c[a]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    dacc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[u][a], bq[u], dacc[a], 0, 0, 0);
        }
        // D[i][j]: i = lk + 4q -> column 16a + i, j = lr -> row 16*cg + lr
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) Ds[(16 * a + lk + 4 * q) * (NB_IN + 1) + rrow] = dacc[a][q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const i32 col = cg + 4 * q;
        // clamped address + select instead of a guarded load: a branch around each load makes it wait
        // for the previous one (16 dependent L2 round trips on the factorisation's serial chain)
        const bool mine = rok && col < nb && r >= col;
        const double pv = P[(i64)min(r, nb - 1) + (i64)min(col, nb - 1) * lda];
        const double dv = (Kp > 0) ? Ds[col * (NB_IN + 1) + r] : 0.0;
        av[q] = mine ? (pv - dv) : 0.0;
        wv[q] = (r == col) ? 1.0 : 0.0;
    }
    // Column steps.  Thread (r, cg) holds columns cg + 4q: column j = 4 jq + jj belongs to the threads with cg == jj, in
    // register av[jq].  The loop over jq is unrolled, so every register index below is static, and a step only touches
    // what can still change: A columns q >= jq (columns before j are final) and W columns q <= jq (row j of the
    // inverse is zero beyond column j) -- 17 multiply-adds, selects and LDS reads per thread and column instead of 32,
    // and no 16-way select chains to extract / update av[jq].
#pragma unroll
    for (int jq = 0; jq < 16; ++jq) {
#pragma unroll 1
        for (int jj = 0; jj < 4; ++jj) {
            const i32 j = 4 * jq + jj;
            if (j >= nb) break;                                  // workgroup-uniform
            const int pb = j & 1;
            if (cg == jj) colbuf[pb][r] = av[jq];                // A[r][j]
            if (r == j) {
#pragma unroll
                for (int q = 0; q <= jq; ++q) rowbuf[pb][cg + 4 * q] = wv[q];   // W[j][0 .. j]
            }
            __syncthreads();
            // every LDS read of the step is issued here, before the pivot arithmetic: the serial chain of a column is one
            // LDS round trip + the reciprocal square root, not three round trips
            double d = colbuf[pb][j];
            const double crj = colbuf[pb][r];                                   // A[r][j] of this thread's row
            double cv[16], rv[16];
#pragma unroll
            for (int q = jq; q < 16; ++q) cv[q] = colbuf[pb][cg + 4 * q];
#pragma unroll
            for (int q = 0; q <= jq; ++q) rv[q] = rowbuf[pb][cg + 4 * q];
            const double sj = SIGNED ? sg[bk0 + j] : 1.0;
            if (!(sj * d > 0.0)) {
                if (tid == 0) atomicMin(c.info, fd.col0 + bk0 + j);
                d = sj;
            }
            if (SIGNED) d = fabs(d);
            // pivot arithmetic off one reciprocal square root (hardware estimate + 2 Newton steps, then a
            // final correction of the square root): isq = |d|^-1/2, sq = |d|^1/2, inv2 = 1/|d| = isq^2
            double isq = __builtin_amdgcn_rsq(d);
            isq = isq * (1.5 - 0.5 * d * isq * isq);
            isq = isq * (1.5 - 0.5 * d * isq * isq);
            double sq = d * isq;
            sq = fma(0.5 * isq, fma(-sq, sq, d), sq);
            const double inv2 = SIGNED ? isq * isq * sj : isq * isq;            // 1 / (signed pivot)
            if (SIGNED) isq *= sj;                                              // column scale s_j / sqrt|d|
            const double arj = (rok && r > j) ? crj * inv2 : 0.0;              // multiplier L~[r][j] (0 for the rows above)
            // Branch-free updates without per-entry masks:
            //  * A[r][col] -= arj * A[col][j] for the columns after j.  Registers q > jq hold only such columns; in
            //    register jq only the threads with cg > jj do.  Entries above the diagonal (col > r) are never read
            //    (they start at 0, stay finite, and are neither broadcast for a row <= their column nor stored);
            //  * W[r][c] -= arj * W[j][c]: row j of the inverse is zero beyond column j, no mask needed.
#pragma unroll
            for (int q = jq + 1; q < 16; ++q) av[q] = fma(-arj, cv[q], av[q]);
            av[jq] = fma(-((cg > jj) ? arj : 0.0), cv[jq], av[jq]);
#pragma unroll
            for (int q = 0; q <= jq; ++q) wv[q] = fma(-arj, rv[q], wv[q]);
            if (cg == jj) av[jq] = (r == j) ? sq : ((r > j) ? av[jq] * isq : av[jq]);   // finish column j: L[r][j]
        }
    }
    // diagonal of L for the row scaling of the inverse
    if (rok && cg == (r & 3)) {
        double v = 1.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) v = (q == (r >> 2)) ? av[q] : v;
        dg[r] = v;
    }
    __syncthreads();
    double *W = front_dinv(c, fd, bk0);          // column-major nb x nb, ld = nb, upper part zero
    if (rok) {
        const double idg = 1.0 / dg[r];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const i32 col = cg + 4 * q;
            if (col < nb) {
                if (r >= col) P[(i64)r + (i64)col * lda] = av[q];
                W[(i64)r + (i64)col * nb] = (r >= col) ? wv[q] * idg : 0.0;   // L^{-1} = diag(1/L_ii) L~^{-1}
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// potrf_block with ONE barrier per TWO columns (round 5).  profiles/r03_potrf_phases.txt: of the ~1050 cycles a column step of potrf_block
// costs, ~485 are the LDS hand-over of the pivot column (write -> barrier -> read).  Here the owners publish the columns j and j + 1 (and the
// rows j, j + 1 of the inverse) AS THEY ARE BEFORE STEP j, and every thread redoes locally what step j would have done to the entries of column
// j + 1 it needs: its own row's A'[r][j+1] = fma(-l_rj, A[j+1][j], A[r][j+1]), the pivot A'[j+1][j+1], the multipliers' column
// A'[col][j+1] for the columns it owns, and row j + 1 of the inverse -- exactly the expressions the owning threads evaluate in potrf_block,
// on the same operands, so the factor and the inverse are BIT-IDENTICAL to potrf_block's (tools/potrf_wave_bench.hip asserts it; the numpy
// model of the pair step: tools/potrf_pair_model.py).  The two broadcast columns are interleaved in LDS ({C0[i], C1[i]} adjacent: one
// 16-byte read fetches a column entry of both), so a pair step issues no more LDS reads than two single steps.  An odd last column
// (j + 1 == nb) runs the same code with the second step masked off.
// ------------------------------------------------------------------------------------------
constexpr int POTRF_PAIR_SCRATCH = NB_IN * (NB_IN + 1) + 8 * NB_IN + NB_IN;   // Ds | CB[2][64][2] | RB[2][64][2] | dg   (Ds starts 16-byte aligned, NB_IN (NB_IN + 1) is even)

template <bool SIGNED = false>
__device__ __forceinline__ void potrf_block_pair(const DevCtx &c, const FrontDesc &fd, const i32 bk0, const i32 nb,
                                                 const i32 kprev, double *scratch) {
    const double *sg = SIGNED ? c.csign + fd.col0 : nullptr;
    double *Ds = scratch;                                   // NB_IN x (NB_IN + 1)
    typedef double v2 __attribute__((ext_vector_type(2)));
    v2 (*CB)[NB_IN] = reinterpret_cast<v2 (*)[NB_IN]>(scratch + NB_IN * (NB_IN + 1));        // CB[pb][i] = {A[i][j], A[i][j+1]} before step j
    v2 (*RB)[NB_IN] = CB + 2;                                                                   // RB[pb][col] = {W[j][col], W[j+1][col]}
    double *dg = scratch + NB_IN * (NB_IN + 1) + 8 * NB_IN;
    const i32 lda = pld(fd, bk0);
    double *P = pcol(c, fd, bk0) + bk0;             // origin (bk0, bk0)
    const int tid = threadIdx.x, r = tid & 63, cg = tid >> 6;
    const bool rok = r < nb;
    double av[16], wv[16];
    // left-looking prologue inside the block column: identical to potrf_block's
    const i32 Kp = bk0 - kprev;
    if (Kp > 0) {
        const int lane = tid & 63, lr = lane & 15, lk = lane >> 4;
        v4f64 dacc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) dacc[a] = (v4f64){0.0, 0.0, 0.0, 0.0};
        const i32 rrow = 16 * cg + lr;
        const i32 rr_c = min(rrow, nb - 1);
        i32 cr_c[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) cr_c[a] = min(16 * a + lr, nb - 1);
        for (i32 ks = 0; ks < Kp; ks += 16) {
            double bq[4], aq[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double *Xc = pcol(c, fd, kprev + ks + 4 * u + lk) + bk0;
                bq[u] = Xc[rr_c];
                if (SIGNED) bq[u] *= sg[kprev + ks + 4 * u + lk];
#pragma unroll
                for (int a = 0; a < 4; ++a) aq[u][a] = Xc[cr_c[a]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    dacc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[u][a], bq[u], dacc[a], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) Ds[(16 * a + lk + 4 * q) * (NB_IN + 1) + rrow] = dacc[a][q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const i32 col = cg + 4 * q;
        const bool mine = rok && col < nb && r >= col;
        const double pv = P[(i64)min(r, nb - 1) + (i64)min(col, nb - 1) * lda];
        const double dv = (Kp > 0) ? Ds[col * (NB_IN + 1) + r] : 0.0;
        av[q] = mine ? (pv - dv) : 0.0;
        wv[q] = (r == col) ? 1.0 : 0.0;
    }
    // pivot of one column from its diagonal entry d (sign s): 1 / (signed pivot), column scale, sqrt|d| -- potrf_block's arithmetic
    auto pivot = [&](double d, const double sj, const i32 j, double &inv2, double &isq, double &sq) {
        if (!(sj * d > 0.0)) {
            if (tid == 0) atomicMin(c.info, fd.col0 + bk0 + j);
            d = sj;
        }
        if (SIGNED) d = fabs(d);
        isq = __builtin_amdgcn_rsq(d);
        isq = isq * (1.5 - 0.5 * d * isq * isq);
        isq = isq * (1.5 - 0.5 * d * isq * isq);
        sq = d * isq;
        sq = fma(0.5 * isq, fma(-sq, sq, d), sq);
        inv2 = SIGNED ? isq * isq * sj : isq * isq;
        if (SIGNED) isq *= sj;
    };
#pragma unroll
    for (int jq = 0; jq < 16; ++jq) {
#pragma unroll 1
        for (int jp = 0; jp < 2; ++jp) {
            const int jj = 2 * jp;
            const i32 j = 4 * jq + jj;
            if (j >= nb) break;                                  // workgroup-uniform
            const bool two = j + 1 < nb;                         // workgroup-uniform: the pair is complete
            const int pb = jp;                                   // consecutive pair steps alternate between the two buffers
            if (cg == jj) CB[pb][r].x = av[jq];                  // A[r][j]
            if (cg == jj + 1) CB[pb][r].y = av[jq];              // A[r][j + 1], not yet updated by column j
            if (r == j) {
#pragma unroll
                for (int q = 0; q <= jq; ++q) RB[pb][cg + 4 * q].x = wv[q];   // W[j][0 .. j]
            }
            if (r == j + 1) {
#pragma unroll
                for (int q = 0; q <= jq; ++q) RB[pb][cg + 4 * q].y = wv[q];   // W[j + 1][0 .. j + 1], not yet updated by step j
            }
            __syncthreads();
            // every LDS read of the pair is issued here, before the pivot arithmetic
            const v2 pj = CB[pb][j], pj1 = CB[pb][j + 1], pr = CB[pb][r];
            v2 cv[16], rv[16];
#pragma unroll
            for (int q = jq; q < 16; ++q) cv[q] = CB[pb][cg + 4 * q];
#pragma unroll
            for (int q = 0; q <= jq; ++q) rv[q] = RB[pb][cg + 4 * q];
            // ---- step j ----
            double inv2_0, isq_0, sq_0;
            pivot(pj.x, SIGNED ? sg[bk0 + j] : 1.0, j, inv2_0, isq_0, sq_0);
            const double arj0 = (rok && r > j) ? pr.x * inv2_0 : 0.0;              // multiplier L~[r][j]
            const double a10 = pj1.x;                                              // A[j + 1][j]
            const double a0_j1 = two ? a10 * inv2_0 : 0.0;                         // multiplier of row j + 1 (row j + 1 < nb iff `two`)
            // ---- what step j does to column j + 1 and to row j + 1 of the inverse, redone locally ----
            const double d1 = fma(-a0_j1, a10, pj1.y);                             // A'[j + 1][j + 1]
            const double c1r = fma(-arj0, a10, pr.y);                              // A'[r][j + 1] of this thread's row
            // ---- step j + 1 ----
            double inv2_1 = 0.0, isq_1 = 0.0, sq_1 = 0.0;
            if (two) pivot(d1, SIGNED ? sg[bk0 + j + 1] : 1.0, j + 1, inv2_1, isq_1, sq_1);
            const double arj1 = (two && rok && r > j + 1) ? c1r * inv2_1 : 0.0;    // multiplier L~[r][j + 1]
            // A[r][col] -= l_rj A[col][j] + l_r,j+1 A'[col][j+1]   (col > j + 1), in potrf_block's order: step j first
#pragma unroll
            for (int q = jq + 1; q < 16; ++q) {
                const i32 col = cg + 4 * q;
                const double a0c = (col < nb) ? cv[q].x * inv2_0 : 0.0;            // multiplier of row `col` at step j (col > j here)
                const double c1c = fma(-a0c, a10, cv[q].y);                        // A'[col][j + 1]
                av[q] = fma(-arj1, c1c, fma(-arj0, cv[q].x, av[q]));
            }
            {
                // register jq: columns 4 jq + cg.  cg <= jj: columns <= j, final or being finished (multipliers masked to 0, as in potrf_block);
                // cg == jj + 1: column j + 1 itself receives step j; cg > jj + 1: both steps
                const i32 col = cg + 4 * jq;
                const double a0c = (col < nb && col > j) ? cv[jq].x * inv2_0 : 0.0;
                const double c1c = fma(-a0c, a10, cv[jq].y);
                av[jq] = fma(-((cg > jj) ? arj0 : 0.0), cv[jq].x, av[jq]);
                av[jq] = fma(-((cg > jj + 1) ? arj1 : 0.0), c1c, av[jq]);
            }
            // W[r][c] -= l_rj W[j][c] + l_r,j+1 W'[j+1][c]
#pragma unroll
            for (int q = 0; q <= jq; ++q) {
                const double r1c = fma(-a0_j1, rv[q].x, rv[q].y);                  // W'[j + 1][c]: row j + 1 after step j
                wv[q] = fma(-arj1, r1c, fma(-arj0, rv[q].x, wv[q]));
            }
            if (cg == jj) av[jq] = (r == j) ? sq_0 : ((r > j) ? av[jq] * isq_0 : av[jq]);                       // finish column j
            if (two && cg == jj + 1) av[jq] = (r == j + 1) ? sq_1 : ((r > j + 1) ? av[jq] * isq_1 : av[jq]);    // finish column j + 1
        }
    }
    // diagonal of L for the row scaling of the inverse
    if (rok && cg == (r & 3)) {
        double v = 1.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) v = (q == (r >> 2)) ? av[q] : v;
        dg[r] = v;
    }
    __syncthreads();
    double *W = front_dinv(c, fd, bk0);
    if (rok) {
        const double idg = 1.0 / dg[r];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const i32 col = cg + 4 * q;
            if (col < nb) {
                if (r >= col) P[(i64)r + (i64)col * lda] = av[q];
                W[(i64)r + (i64)col * nb] = (r >= col) ? wv[q] * idg : 0.0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// potrf_block without a barrier per column (round 4; tools/potrf64_probe.hip, profiles/r03_potrf_phases.txt): the 64-column loop of potrf_block
// costs ~1050 cycles per column, of which ~485 are the LDS hand-over of the pivot column (write -> barrier -> read) and ~280 the LDS
// bandwidth of four waves' broadcast reads.  Here ONE wave factors the block: lane r owns row r, the block is walked in four 16-column
// panels held in registers, every broadcast inside a panel is a v_readlane (no LDS round trip, no barrier), the left-looking update of a
// panel by the previous panels and the off-diagonal blocks of the inverse run on the matrix cores with operands in LDS:
//     panel p:   A[:, p] -= L[:, <p] S L[p, <p]'        (MFMA, operands from the transposed copy Mt of the finished panels)
//                16 column steps on [A_pp | I]           (readlane broadcasts; gives L[:, p] and W_pp = L_pp^-1)
//     inverse:   W_ij = -W_ii sum_{k=j}^{i-1} L_ik W_kj   by block distance (MFMA)
// One LDS array holds both factors: Mt[x][y] = L[row y][column x] in the blocks BELOW the diagonal blocks (y-block > x-block) and
// W[row x][column y] in the blocks on and below them (x-block >= y-block) -- the matrix-core products never read a diagonal block of L.
// The other three waves of the workgroup wait at the barrier that follows (they join in again for the rows below the block, trsm_rows).
// Same results as potrf_block up to rounding (same eliminations, different summation order in the panel updates); a pivot of the wrong sign
// is reported and replaced in the same way.  Partial blocks (nb < 64): rows >= nb are rows of the identity.
// MEASURED (round 4, tools/potrf_wave_bench.hip, profiles/r04_potrf_wave.txt): parity-green (K1, K2, partial blocks, failure reporting), and SLOWER
// than potrf_block in the kernels -- 43.9 vs 34.4 us per 64 x 64 block, k_potrf_wide 230 vs 196 us per 256-wide block, potrf class on config C4
// 4.9 vs 3.8 ms: the 16 column steps of a panel take 4.3 us (270 ns per column against 530 ns in potrf_block, as the probe promised), but the
// rest of a panel -- 48 predicated stores with their address arithmetic, the LDS copies the matrix-core products need, the serial panel updates
// -- costs another 5 us on the same single wave, and the probe's 21 us had left exactly that out.  Kept behind TLPK_POTRF_WAVE=1 (default: potrf_block).
// ------------------------------------------------------------------------------------------
constexpr int PW_W = 16;                          // panel width
constexpr int PW_LDT = 17;                        // leading dimension of the small transposition arrays
constexpr int LDW_ = NB_IN + 16;                  // == LDW (declared below): == 16 mod 32, conflict-free ds_read_b64 of the MFMA operands
constexpr int POTRF_WAVE_LDS = NB_IN * LDW_ + 4 * PW_W * PW_LDT + NB_IN * PW_LDT;      // doubles: Mt | Wd[4] | Ts
#define TLPK_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <bool SIGNED = false>
__device__ __forceinline__ void potrf_block_wave(const DevCtx &c, const FrontDesc &fd, const i32 bk0, const i32 nb,
                                                 const i32 kprev, double *scratch) {
    const double *sg = SIGNED ? c.csign + fd.col0 : nullptr;
    double *Mt = scratch;                                        // NB_IN x LDW_   (first: Ds, then As = the block to factor, row for row)
    double *Wd = scratch + NB_IN * LDW_;                         // Wd[p][t * PW_LDT + r] = W_pp[r][t]  (column-major copies of the diagonal blocks)
    double *Ts = Wd + 4 * PW_W * PW_LDT;                         // transposition scratch, one NB_IN/4 x PW_LDT piece per wave in the inverse phase
    double *Ds = scratch;
    const i32 lda = pld(fd, bk0);
    double *P = pcol(c, fd, bk0) + bk0;                          // origin (bk0, bk0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
#ifdef POTRF_TRACE   /* tools/potrf_wave_bench.hip: 100 MHz stamps of the phases, one row of 32 per workgroup in c.spart */
    unsigned long long *ptr_ = (unsigned long long *)c.spart + (size_t)blockIdx.x * 32;
    int pti_ = 0;
#define PT_STAMP() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (tid == 0 && pti_ < 32) ptr_[pti_] = wall_clock64(); ++pti_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PT_STAMP() do {} while (0)
#endif
    PT_STAMP();
    // the block itself, requested by all 256 threads before anything else (thread (r, cg): columns cg + 4 q; clamped addresses, selects afterwards)
    const int r = lane;
    const bool rok = r < nb;
    double pv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) pv[q] = P[(i64)min(r, nb - 1) + (i64)min(wave + 4 * q, nb - 1) * lda];
    // left-looking over the already factored 64-wide steps of this block column (all four waves; as in potrf_block)
    const i32 Kp = bk0 - kprev;
    if (Kp > 0) {
        v4f64 dacc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) dacc[a] = (v4f64){0.0, 0.0, 0.0, 0.0};
        const i32 rrow = 16 * wave + lr;
        const i32 rr_c = min(rrow, nb - 1);
        i32 cr_c[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) cr_c[a] = min(16 * a + lr, nb - 1);
        for (i32 ks = 0; ks < Kp; ks += 16) {
            double bq[4], aq[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double *Xc = pcol(c, fd, kprev + ks + 4 * u + lk) + bk0;
                bq[u] = Xc[rr_c];
                if (SIGNED) bq[u] *= sg[kprev + ks + 4 * u + lk];
#pragma unroll
                for (int a = 0; a < 4; ++a) aq[u][a] = Xc[cr_c[a]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    dacc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[u][a], bq[u], dacc[a], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) Ds[(16 * a + lk + 4 * q) * LDW_ + rrow] = dacc[a][q];
        __syncthreads();
    }
    // As = block - Ds, in place of Ds (same stride, same rows); rows / columns beyond nb: identity
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int col = wave + 4 * q;
        const double dv = (Kp > 0) ? Ds[col * LDW_ + r] : 0.0;
        Mt[col * LDW_ + r] = (rok && col < nb) ? ((r >= col) ? (pv[q] - dv) : 0.0) : ((r == col) ? 1.0 : 0.0);
    }
    __syncthreads();
    PT_STAMP();
    double *W = front_dinv(c, fd, bk0);                          // column-major nb x nb, ld = nb, upper part zero
    if (wave == 0) {
    // The chain: no global loads from here on (the stores are fire-and-forget: a load in between would make every panel wait for the stores of
    // the previous one).  Panel p reads rows [16 p, 16 p + 16) of As before it writes those rows of Mt.
    i32 failcol = NB_IN;                                         // first pivot of the wrong sign (NB_IN: none)
    // (the panel loop is NOT unrolled: four copies of the 16 column steps next to trsm_rows in k_potrf_wide pushed the kernel past 256
    // registers into scratch)
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
        double a[PW_W], w[PW_W];
#pragma unroll
        for (int cc = 0; cc < PW_W; ++cc) { a[cc] = Mt[(PW_W * p + cc) * LDW_ + r]; w[cc] = (r == PW_W * p + cc) ? 1.0 : 0.0; }
        TLPK_LDS_FENCE();                                        // this panel's rows of As are consumed
        if (p > 0) {
            // U[row][cc] = sum_{k < 16 p} L[row][k] s_k L[16 p + cc][k] for the row blocks b >= p
#pragma unroll 1
            for (int b = p; b < 4; ++b) {
                v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
                for (int k4 = 0; k4 < PW_W * p; k4 += 4) {
                    double x = Mt[(k4 + lk) * LDW_ + PW_W * p + lr];
                    if (SIGNED) x *= sg[bk0 + min(k4 + lk, nb - 1)];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, Mt[(k4 + lk) * LDW_ + PW_W * b + lr], acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) Ts[(PW_W * b + lr) * PW_LDT + lk + 4 * q] = acc[q];
            }
            TLPK_LDS_FENCE();
#pragma unroll
            for (int cc = 0; cc < PW_W; ++cc) a[cc] -= (r >= PW_W * p) ? Ts[r * PW_LDT + cc] : 0.0;
            TLPK_LDS_FENCE();
        }
        PT_STAMP();
#pragma unroll
        for (int j = 0; j < PW_W; ++j) {
            const int J = PW_W * p + j;
            double d = readlane_f64(a[j], J);
            const double sj = (SIGNED && J < nb) ? sg[bk0 + J] : 1.0;
            const bool bad = !(sj * d > 0.0);                                   // (wave-uniform) reported once, after the block: no branch, no atomic on the chain
            failcol = bad ? min(failcol, J) : failcol;
            d = bad ? sj : d;
            if (SIGNED) d = fabs(d);
            double isq = __builtin_amdgcn_rsq(d);
            isq = isq * (1.5 - 0.5 * d * isq * isq);
            isq = isq * (1.5 - 0.5 * d * isq * isq);
            double sq = d * isq;
            sq = fma(0.5 * isq, fma(-sq, sq, d), sq);
            const double inv2 = SIGNED ? isq * isq * sj : isq * isq;            // 1 / (signed pivot)
            if (SIGNED) isq *= sj;                                              // column scale s_j / sqrt|d|
            // (the row's position relative to the panel goes through an empty asm in every step: the 32 lane masks r > J, r == J of a panel are
            // otherwise hoisted out of the column loop into 64 scalar registers and spilled through v_writelane)
            int rq = r - PW_W * p;
            asm volatile("" : "+v"(rq));
            const bool below = rq > j, diag = rq == j;
            const double arj = below ? a[j] * inv2 : 0.0;
#pragma unroll
            for (int cc = j + 1; cc < PW_W; ++cc) a[cc] = fma(-arj, readlane_f64(a[j], PW_W * p + cc), a[cc]);
#pragma unroll
            for (int cc = 0; cc <= j; ++cc) w[cc] = fma(-arj, readlane_f64(w[cc], J), w[cc]);
            a[j] = diag ? sq : (below ? a[j] * isq : a[j]);
            __builtin_amdgcn_sched_barrier(0);          // keep the broadcasts of a step inside the step: hoisted across steps they overflow the scalar registers (spilled through v_writelane)
        }
        PT_STAMP();
        double lii = 1.0;
#pragma unroll
        for (int cc = 0; cc < PW_W; ++cc) lii = (r == PW_W * p + cc) ? a[cc] : lii;
        const double ili = 1.0 / lii;
        const bool inblk = (r >= PW_W * p) && (r < PW_W * p + PW_W);
#pragma unroll
        for (int cc = 0; cc < PW_W; ++cc) {
            const int col = PW_W * p + cc;
            const double lv = (r >= col) ? a[cc] : 0.0;
            if (r >= PW_W * (p + 1)) Mt[col * LDW_ + r] = lv;                    // L below the diagonal blocks, transposed
            if (rok && col < nb && r >= col) P[(i64)r + (i64)col * lda] = lv;
            if (inblk) {
                const double wv = (r >= col) ? w[cc] * ili : 0.0;
                Mt[r * LDW_ + col] = wv;                                         // W, row-major, diagonal block (zeros above the diagonal)
                Wd[p * PW_W * PW_LDT + cc * PW_LDT + (r - PW_W * p)] = wv;
                if (rok && col < nb) W[(i64)r + (i64)col * nb] = wv;
            } else if (rok && col < nb && r < PW_W * p) W[(i64)r + (i64)col * nb] = 0.0;      // blocks above the diagonal blocks: zero, as potrf_block leaves them
        }
        TLPK_LDS_FENCE();
    }
    if (failcol < NB_IN && r == 0) atomicMin(c.info, fd.col0 + bk0 + failcol);
    }
    PT_STAMP();
    // off-diagonal blocks of the inverse, by block distance: W_ij = -W_ii (sum_{k=j}^{i-1} L_ik W_kj); the blocks of one distance are independent:
    // one wave each, a barrier between the distances
#pragma unroll 1
    for (int dist = 1; dist < 4; ++dist) {
        __syncthreads();
        const int i = dist + wave, j = wave;
        if (i < 4) {                                            // (wave-uniform)
            double *Tw = Ts + wave * (PW_W * PW_LDT);
            // G[rr][cc] = sum_k sum_t L_ik[rr][t] W_kj[t][cc]: first operand M1[cc][t] = W_kj[t][cc], second M2[rr][t] = L_ik[rr][t]
            v4f64 g = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
            for (int k = j; k < i; ++k) {
#pragma unroll
                for (int k4 = 0; k4 < PW_W; k4 += 4)
                    g = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[(PW_W * k + k4 + lk) * LDW_ + PW_W * j + lr], Mt[(PW_W * k + k4 + lk) * LDW_ + PW_W * i + lr], g, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) Tw[lr * PW_LDT + lk + 4 * q] = g[q];      // G[rr = lr][cc = lk + 4q] -> Tw[t = rr][cc]
            TLPK_LDS_FENCE();
            // H[rr][cc] = sum_t W_ii[rr][t] G[t][cc]: first operand M1[cc][t] = G[t][cc], second M2[rr][t] = W_ii[rr][t]
            v4f64 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k4 = 0; k4 < PW_W; k4 += 4)
                h = __builtin_amdgcn_mfma_f64_16x16x4f64(Tw[(k4 + lk) * PW_LDT + lr], Wd[i * PW_W * PW_LDT + (k4 + lk) * PW_LDT + lr], h, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = PW_W * i + lr, cc = PW_W * j + lk + 4 * q;
                Mt[rr * LDW_ + cc] = -h[q];
                if (rr < nb && cc < nb) W[(i64)rr + (i64)cc * nb] = -h[q];
            }
        }
    }
    PT_STAMP();
}

// ------------------------------------------------------------------------------------------
// potrf_block on DPP broadcasts (round 5).  The 64 x 64 block is walked in four 16-column panels as in potrf_block_wave, but the
// 16 column steps of a panel run in the layout  lane (g, cc) = column cc of the panel, registers = rows:  d[k] = A[16 p + k][16 p + cc] (the 16 x 16
// diagonal block, replicated in the wave's four rows of 16 lanes; the lane of column c WORKS on the entries (k, c), k < c -- what its registers hold
// behind the diagonal is never read) and b[k] = a row of the panel below the diagonal block (the 16 (3 - p) rows are dealt to the four rows of lanes:
// 4 (3 - p) registers).  The multipliers of step j are ROW j of the triangle: A[j][c] is the lane's own register j, A[j][k] is lane k's register j, so
// with w = -A[j][c] / d_j formed once per lane an entry is updated by ONE instruction,
//     v_fmac_f64_dpp d[k], w, d[j] row_newbcast:k      (A[k][c] += w(lane k) * A[j][c])         diagonal block
//     v_fmac_f64_dpp b[k], b[k], w row_newbcast:j      (A[r][c] += A[r][j](lane j) * w)          rows below
// -- `row_newbcast` (lane j of each row of 16 lanes) is the DPP form CDNA has for 64-bit operands (v_fmac_f64, v_mov_b64 only).  No v_readlane -> scalar
// register -> multiply-add round trips (potrf_block_wave: 456 cycles per column), no LDS hand-over and no barrier (potrf_block: ~1050 cycles); the
// serial chain of a step is the broadcast of the next pivot, a reciprocal (hardware estimate + 2 Newton steps), the multiplier and the multiply-add of
// the next pivot: 85 - 125 ns per column.  Column scaling is deferred: the registers keep the unscaled Schur state, every lane remembers ITS column's
// pivot and scales once after the 16 steps.  (Two earlier versions kept both triangles of the diagonal block: see dpp_steps.)
// The DPP instructions are inline assembly (the compiler splits a 64-bit DPP move off every multiply-add and then waits a cycle for its own
// temporary); the statements keep the two-instruction distance the hardware requires between a write of a register and a DPP read of it (the compiler's
// hazard recogniser does not look inside inline assembly): an s_nop in front, or the order of the instructions inside the statement.
// Between panels the trailing 16 x 16 tiles are updated right-looking on the matrix cores (K = 16): the tiles the next panel reads by waves 1..3
// between two barriers.  WHILE wave 0 runs the next panel, wave 1 inverts the finished diagonal block (forward substitution in the same DPP layout),
// waves 2, 3 update the other tiles, store the finished panel and form the sums G_ij = sum_k L_ik W_kj of the inverse's off-diagonal blocks whose
// operands are older than the last barrier; W_ij = -W_ii G_ij follows in the next phase that has W_ii (one barrier is left after W_33).
// LDS: Mt as in potrf_block_wave (the part of the block not yet factored in place of L), leading dimension PD_LD = 82 (lane stride 164 dwords = 36 mod
// 64 banks: the 16-byte accesses of wave 0, one column per lane, are conflict-free).
// A pivot of the wrong sign is reported as in potrf_block; the panel that holds it and the panels after it become columns of the identity
// (nothing non-finite is stored).  Partial blocks (nb < 64): rows / columns >= nb are rows of the identity.
// ------------------------------------------------------------------------------------------
constexpr int PD_LD = NB_IN + 18;                  // leading dimension of Mt in potrf_block_dpp
constexpr int PD_LP = PW_W * PW_W;                 // L_pp, column-major, for the forward substitution and the stores
constexpr int POTRF_DPP_LDS = NB_IN * PD_LD + 4 * PW_W * PW_LDT + 4 * PW_W * PW_LDT + 2 * PD_LP + NB_IN;  // doubles: Mt | Wd[4] | Ts[4] | Lp[2] | Sg
constexpr int TRM_LDT = NB_IN + 2;                 // trsm_rows_mt: leading dimension of the shared block of L behind Mt (== 2 mod 32: conflict-free b64 operand reads)
constexpr int POTRF_WIDE_DPP_LDS = NB_IN * PD_LD + NB_IN * TRM_LDT;                 // doubles: Mt | Wt (>= POTRF_DPP_LDS)
static_assert(30 * 256 <= POTRF_DPP_LDS, "potrf_block_dpp: the waves' shares of the left-looking sum (30 slots of 256 doubles) lie in the image's place");
static_assert(POTRF_WIDE_DPP_LDS >= POTRF_DPP_LDS, "k_potrf_wide: the in-block solve's block of L lies behind the diagonal block's image");
#ifndef TLPK_PROLOGUE_KSPLIT
#define TLPK_PROLOGUE_KSPLIT 1
#endif
#ifndef TLPK_TRM_MT
#define TLPK_TRM_MT 1
#endif
constexpr bool TRM_MT = TLPK_TRM_MT != 0;         // 0 (build-time, diagnostics): the round-5 in-block solves (trsm_rows: blocks staged one global round trip at a time)

template <int J>
__device__ __forceinline__ double bc16(const double x) {         // lane J of the caller's row of 16 lanes
    return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + J, 0xf, 0xf, true);
}

#define TLPK_DPP_FMAC(i) "v_fmac_f64_dpp %" #i ", %" #i ", %[nm] row_newbcast:%c[j] row_mask:0xf bank_mask:0xf\n"
#define TLPK_DPP_FMK(i) "v_fmac_f64_dpp %" #i ", %[w], %[v] row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n"
#define TLPK_DPP_EARLY(i) ".if (" #i " > %c[j]) && (" #i " <= %c[j] + 3)\n" TLPK_DPP_FMK(i) ".endif\n"
#define TLPK_DPP_LATE(i) ".if " #i " > %c[j] + 3\n" TLPK_DPP_FMK(i) ".endif\n"
#define TLPK_DPP_NEXT(i) ".if " #i " == %c[j] + 1\nv_mov_b64_dpp %0, %" #i " row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n.endif\n"
#define TLPK_DPP_ALL15(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
// step J on the diagonal block: d[k] += w(lane k) * v for the registers k = J + 1 .. 15 -- entry (k, c) of the block, held by the lane of column c,
// takes -A[j][k] / d_j from lane k (w, read through `row_newbcast:k`) and A[j][c] from its own register j (v) -- and dn = lane J + 1's d[J + 1],
// the next pivot, two instructions after that register was written.  (s_nop 1: w was written by the instruction in front of the statement.)
template <int J>
__device__ __forceinline__ void dpp_dblock(double (&d)[PW_W], const double w, const double v, double &dn) {
    static_assert(J < PW_W - 1, "the last step has no trailing part");
    asm("s_nop 1\n"
        TLPK_DPP_ALL15(TLPK_DPP_EARLY)
        ".if %c[j] == 13\ns_nop 0\n.endif\n.if %c[j] == 14\ns_nop 1\n.endif\n"
        TLPK_DPP_ALL15(TLPK_DPP_NEXT)
        TLPK_DPP_ALL15(TLPK_DPP_LATE)
        : "=&v"(dn), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]), "+v"(d[10]),
          "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15])
        : [w] "v"(w), [v] "v"(v), [j] "n"(J));
}
#undef TLPK_DPP_ALL15
#undef TLPK_DPP_NEXT
#undef TLPK_DPP_LATE
#undef TLPK_DPP_EARLY
#undef TLPK_DPP_FMK
// four rows below the diagonal block: b += bcast_J(b) * nm (the s_nop: the registers were written by the previous step's block, which the
// compiler may have placed right in front of this one)
template <int J>
__device__ __forceinline__ void dpp_bchunk(double &b0, double &b1, double &b2, double &b3, const double nm) {
    asm("s_nop 1\n" TLPK_DPP_FMAC(0) TLPK_DPP_FMAC(1) TLPK_DPP_FMAC(2) TLPK_DPP_FMAC(3)
        : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : [nm] "v"(nm), [j] "n"(J));
}
// the two Newton steps of step J's reciprocal (x = rcp(dj) on entry) with the multiply-adds of step JP = J - 1 on the NBR rows below the diagonal
// block (multiplier nmp) in the issue slots behind each dependent operation.  (s_nop 1: the trans -> VALU distance of x, and the DPP distance of b.)
#define TLPK_DPP_BIF(i, n) ".if %c[nbr] > " #n "\n" TLPK_DPP_FMAC(i) ".endif\n"
template <int JP, int NBR>
__device__ __forceinline__ void dpp_newton_b(double &x, const double dj, double (&b)[12], const double nmp) {
    double e;
    asm("s_nop 1\n"
        "v_fma_f64 %[e], -%[dj], %[x], 1.0\n" TLPK_DPP_BIF(0, 0)
        "v_fmac_f64_e32 %[x], %[x], %[e]\n" TLPK_DPP_BIF(1, 1) TLPK_DPP_BIF(2, 2)
        "v_fma_f64 %[e], -%[dj], %[x], 1.0\n" TLPK_DPP_BIF(3, 3) TLPK_DPP_BIF(4, 4)
        "v_fmac_f64_e32 %[x], %[x], %[e]\n" TLPK_DPP_BIF(5, 5) TLPK_DPP_BIF(6, 6) TLPK_DPP_BIF(7, 7) TLPK_DPP_BIF(8, 8) TLPK_DPP_BIF(9, 9)
        TLPK_DPP_BIF(10, 10) TLPK_DPP_BIF(11, 11)
        : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]),
          [x] "+v"(x), [e] "=&v"(e)
        : [dj] "v"(dj), [nm] "v"(nmp), [j] "n"(JP), [nbr] "n"(NBR));
}
#undef TLPK_DPP_BIF
#undef TLPK_DPP_FMAC

// column steps J .. 15 of one panel (template recursion: the DPP lane selects are immediates).  dj = the pivot of step J (same in every lane).
// Diagonal block: the lane of column c works on the entries (k, c) with k < c -- its registers k < c; the registers behind them are never read --
// so the multipliers of step j are ROW j of that triangle: A[j][c] is the lane's own register j, A[j][k] is lane k's register j, and the factor
// column j is what the lanes k > j hold in register j.  Every entry is updated from those numbers only: the standard elimination on one triangle
// (a first version kept both triangles and took one factor from each: the two copies differ in the last bit, and on the quasi-definite system,
// whose multipliers reach 1e8, the device-resident loops then needed 11 - 19 iterations instead of 8 - 14).
// The rows below the diagonal block take the multiplier form with lane j's registers as the pivot column, and lag one step behind: their
// multiply-adds of step J - 1 (multiplier nmp) stand between the dependent operations of step J's reciprocal, where the wave would otherwise wait.
template <int NBR, int J>
__device__ __forceinline__ void dpp_steps(double (&d)[PW_W], double (&b)[12], const double dj, const double nmp, const double (&gt)[PW_W], const int cc,
                                          double &pv_own) {
    double x = __builtin_amdgcn_rcp(dj);                            // 1 / (signed pivot)
    const double dm = d[J] * gt[J];                                 // A[j][c] in the lanes of the columns c > j, 0 in the finished ones
    if constexpr (J > 0 && NBR > 0) dpp_newton_b<J - 1, NBR>(x, dj, b, nmp);
    else {
        double e = fma(-dj, x, 1.0);
        x = fma(x, e, x);
        e = fma(-dj, x, 1.0);
        x = fma(x, e, x);
    }
    const double nm = -(dm * x);                                    // -A[j][c] / d_j
    int cj = cc;
    asm volatile("" : "+v"(cj));                                    // keeps the lane masks of a panel's 16 steps out of the scalar registers
    pv_own = (cj == J) ? dj : pv_own;
    if constexpr (J + 1 < PW_W) {
        double dn;
        dpp_dblock<J>(d, nm, d[J], dn);
        dpp_steps<NBR, J + 1>(d, b, dn, nm, gt, cc, pv_own);
    } else {
        if constexpr (NBR >= 4) dpp_bchunk<J>(b[0], b[1], b[2], b[3], nm);
        if constexpr (NBR >= 8) dpp_bchunk<J>(b[4], b[5], b[6], b[7], nm);
        if constexpr (NBR >= 12) dpp_bchunk<J>(b[8], b[9], b[10], b[11], nm);
    }
}

// d[j] *= (the scale of column j, held by lane j), j = J .. 15
template <int J>
__device__ __forceinline__ void dpp_scale_rows(double (&d)[PW_W], const double isq) {
    d[J] *= bc16<J>(isq);
    if constexpr (J + 1 < PW_W) dpp_scale_rows<J + 1>(d, isq);
}

// panel P of the block (wave 0): 16 column steps on the registers, then the scaled columns go to LDS (Lp: the diagonal block, Mt: the rows below)
template <bool SIGNED, int P>
__device__ __forceinline__ void dpp_panel(double *Mt, double *Lp, const double *Sg, const int lane_, i32 &failcol) {
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    constexpr int NBR = 4 * (3 - P);                                // registers of rows below the diagonal block per lane
    int lane = lane_;
    asm volatile("" : "+v"(lane));                                  // (nothing derived from the lane index is hoisted out of the panel loop: register pressure of the other waves' code)
    const int g = lane >> 4, cc = lane & 15;
    double d[PW_W], b[12], gt[PW_W];                                // gt[j] = 1 in the lanes whose column lies behind column j of the panel
#pragma unroll
    for (int j = 0; j < PW_W; ++j) gt[j] = (cc > j) ? 1.0 : 0.0;
    double *Cd = Mt + (PW_W * P + cc) * PD_LD;                      // column 16 P + cc of the block
#pragma unroll
    for (int k = 0; k < PW_W; k += 2) {
        const v2f64 t = *reinterpret_cast<const v2f64 *>(Cd + PW_W * P + k);
        d[k] = t.x; d[k + 1] = t.y;
    }
#pragma unroll
    for (int k = 0; k < 12; k += 2) {
        if (k < NBR) {
            const v2f64 t = *reinterpret_cast<const v2f64 *>(Cd + PW_W * (P + 1) + g * NBR + k);
            b[k] = t.x; b[k + 1] = t.y;
        } else { b[k] = 0.0; b[k + 1] = 0.0; }
    }
    double pv_own = 1.0;
    dpp_steps<NBR, 0>(d, b, bc16<0>(d[0]), 0.0, gt, cc, pv_own);
    const double sgl = SIGNED ? Sg[PW_W * P + cc] : 1.0;
    const unsigned long long badm = __builtin_amdgcn_ballot_w64(!(sgl * pv_own > 0.0)) & 0xffffull;      // (every row of lanes holds the same pivots)
    if (badm && failcol == NB_IN) failcol = PW_W * P + __builtin_ctzll(badm);
    if (failcol < NB_IN) {                                          // (wave-uniform) this panel or an earlier one failed: columns of the identity
#pragma unroll
        for (int k = 0; k < PW_W; ++k) d[k] = (k == cc) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 12; ++k) b[k] = 0.0;
    } else {
        const double ad = SIGNED ? fabs(pv_own) : pv_own;
        double isq = __builtin_amdgcn_rsq(ad);
        isq = isq * (1.5 - 0.5 * ad * isq * isq);
        isq = isq * (1.5 - 0.5 * ad * isq * isq);
        if (SIGNED) isq *= sgl;                                     // column scale s_j / sqrt|d|; the diagonal entry d * s / sqrt|d| = sqrt|d|
        dpp_scale_rows<0>(d, isq);                                  // lane r, register j <= r: L[r][j] = A[r][j] * (scale of column j)
#pragma unroll
        for (int k = 0; k < 12; ++k) b[k] *= isq;
    }
    if (g == 0) {                                                   // lane r holds ROW r of the diagonal block (the registers behind the diagonal: never read)
#pragma unroll
        for (int j = 0; j < PW_W; ++j) Lp[(P & 1) * PD_LP + j * PW_W + cc] = d[j];
    }
#pragma unroll
    for (int k = 0; k < 12; k += 2)
        if (k < NBR) *reinterpret_cast<v2f64 *>(Cd + PW_W * (P + 1) + g * NBR + k) = (v2f64){b[k], b[k + 1]};
}

// x = L_pp^-1 e_cc for the lane's column cc, from l[k] = L_pp[k][cc]: steps J .. 15 of the forward substitution, x[k] -= L[k][J] x[J] with
// L[k][J] = lane J's l[k] (one v_fmac_f64_dpp each; l is never written here, the s_nop covers whatever the compiler did to it before)
#define TLPK_DPP_XK(i) ".if " #i " > %c[j]\nv_fmac_f64_dpp %[x" #i "], %[l" #i "], %[nx] row_newbcast:%c[j] row_mask:0xf bank_mask:0xf\n.endif\n"
template <int J>
__device__ __forceinline__ void dpp_inv_steps(const double (&l)[PW_W], double (&x)[PW_W], const double il) {
    x[J] *= bc16<J>(il);
    if constexpr (J + 1 < PW_W) {
        const double nx = -x[J];
        asm("s_nop 1\n" TLPK_DPP_XK(1) TLPK_DPP_XK(2) TLPK_DPP_XK(3) TLPK_DPP_XK(4) TLPK_DPP_XK(5) TLPK_DPP_XK(6) TLPK_DPP_XK(7) TLPK_DPP_XK(8)
            : [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]), [x6] "+v"(x[6]), [x7] "+v"(x[7]), [x8] "+v"(x[8])
            : [l1] "v"(l[1]), [l2] "v"(l[2]), [l3] "v"(l[3]), [l4] "v"(l[4]), [l5] "v"(l[5]), [l6] "v"(l[6]), [l7] "v"(l[7]), [l8] "v"(l[8]), [nx] "v"(nx), [j] "n"(J));
        asm("s_nop 1\n" TLPK_DPP_XK(9) TLPK_DPP_XK(10) TLPK_DPP_XK(11) TLPK_DPP_XK(12) TLPK_DPP_XK(13) TLPK_DPP_XK(14) TLPK_DPP_XK(15)
            : [x9] "+v"(x[9]), [x10] "+v"(x[10]), [x11] "+v"(x[11]), [x12] "+v"(x[12]), [x13] "+v"(x[13]), [x14] "+v"(x[14]), [x15] "+v"(x[15])
            : [l9] "v"(l[9]), [l10] "v"(l[10]), [l11] "v"(l[11]), [l12] "v"(l[12]), [l13] "v"(l[13]), [l14] "v"(l[14]), [l15] "v"(l[15]), [nx] "v"(nx), [j] "n"(J));
        dpp_inv_steps<J + 1>(l, x, il);
    }
}
#undef TLPK_DPP_XK

template <bool SIGNED = false>
__device__ __forceinline__ void potrf_block_dpp(const DevCtx &c, const FrontDesc &fd, const i32 bk0, const i32 nb,
                                                const i32 kprev, double *scratch, unsigned *prog = nullptr) {
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    const double *sg = SIGNED ? c.csign + fd.col0 : nullptr;
    double *Mt = scratch;                                        // NB_IN x PD_LD
    double *Wd = scratch + NB_IN * PD_LD;                        // Wd[p][t * PW_LDT + r] = W_pp[r][t]
    double *Ts = Wd + 4 * PW_W * PW_LDT;                         // one PW_W x PW_LDT transposition piece per wave
    double *Lp = Ts + 4 * PW_W * PW_LDT;                         // Lp[p & 1][t * PW_W + r] = L_pp[r][t] (two buffers: waves 1..3 read panel p - 1's while wave 0 writes panel p's)
    double *Sg = Lp + 2 * PD_LP;                                 // column signs of the block (SIGNED)
    double *Ds = scratch;
    const i32 lda = pld(fd, bk0);
    double *P = pcol(c, fd, bk0) + bk0;                          // origin (bk0, bk0)
    double *W = front_dinv(c, fd, bk0);                          // column-major nb x nb, ld = nb, upper part zero
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int r = lane;
    const bool rok = r < nb;
    const bool wfull = nb == NB_IN && !((size_t)W & 15);         // (workgroup-uniform) 16-byte stores of the inverse
#ifdef POTRF_TRACE   /* tools/potrf_wave_bench.hip: 100 MHz stamps of the phases, one row of 32 per workgroup in c.spart */
    unsigned long long *ptr_ = (unsigned long long *)c.spart + (size_t)blockIdx.x * 32;
    int pti_ = 0;
#endif
    PT_STAMP();
#if TLPK_PROLOGUE_KSPLIT
    // Round 6 (last): the block's own entries and its left-looking sum in the MATRIX-CORE layout, the sum split over the waves by COLUMNS.  The ten 16 x 16 blocks on
    // and below the diagonal belong to the waves 3 + 3 + 2 + 2 (tables below); thread (lr, lk) of the owner holds the entries (16 bi + lr, 16 bj + lk + 4 q).
    // Left-looking part, D -= X S X' over the 64-wide steps already factored in this block column: wave w takes the columns 16 w .. 16 w + 15 of EVERY step and forms
    // its share of ALL ten blocks straight from global memory -- lane (lr, lk) loads L[bk0 + 16 a + lr][kprev + 64 j + 16 w + 4 g + lk], which is the operand layout of
    // v_mfma_f64_16x16x4_f64 for both sides of the product -- so a step needs no staging block in LDS and no barrier (the staged form: two barriers, 16 LDS writes and
    // 32 LDS reads per block and step around 48 products; 2.9 us per step for 1.3 us of matrix-core time).  The shares meet once, in LDS: a wave writes the blocks it
    // does not own, the owner adds the four shares in wave order (fixed: deterministic; both triangles of a diagonal block stay bitwise equal, the products commute).
    // Other summation order than the staged form (columns in ascending order there): results differ in the last bits, like any two of the block kernels.
    const i32 Kp = bk0 - kprev;
    const int nblk = wave < 2 ? 3 : 2, bfirst = wave < 2 ? 3 * wave : 2 * wave + 2;      // blocks 0-2 | 3-5 | 6-7 | 8-9 of the table
    int bi[3], bj[3];                                            // wave 0: (0,0) (3,0) (3,1) | 1: (1,0) (1,1) (3,2) | 2: (2,0) (2,1) | 3: (2,2) (3,3)
    bi[0] = wave == 3 ? 2 : wave; bj[0] = wave == 3 ? 2 : 0;
    bi[1] = wave < 2 ? (wave == 0 ? 3 : 1) : (wave == 2 ? 2 : 3); bj[1] = wave == 0 ? 0 : (wave == 3 ? 3 : 1);
    bi[2] = 3; bj[2] = wave == 0 ? 1 : 2;
    v4f64 tot[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) tot[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    double pvt[3][4];
    if (Kp > 0) {
        constexpr int BI[10] = {0, 3, 3, 1, 1, 3, 2, 2, 2, 3}, BJ[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3}, OW[10] = {0, 0, 0, 1, 1, 1, 2, 2, 3, 3};
        const int nprev = Kp / NB_IN;                            // 1 .. 3 (workgroup-uniform)
        double x[3][4][4];
        i32 rowa[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) rowa[a] = bk0 + min(16 * a + lr, nb - 1);     // clamped: the rows >= nb feed entries nobody reads
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nprev) {
                const double *Pc = pcol(c, fd, kprev + j * NB_IN);
                const i32 ldc = pld(fd, kprev + j * NB_IN);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int a = 0; a < 4; ++a) x[j][a][g] = Pc[(i64)rowa[a] + (i64)(16 * wave + 4 * g + lk) * ldc];
            }
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < nblk) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pvt[t][q] = P[(i64)min(16 * bi[t] + lr, nb - 1) + (i64)min(16 * bj[t] + lk + 4 * q, nb - 1) * lda];
            }
        v4f64 dacc[10];
#pragma unroll
        for (int b = 0; b < 10; ++b) dacc[b] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nprev) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    double xs[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) xs[a] = SIGNED ? x[j][a][g] * sg[kprev + j * NB_IN + 16 * wave + 4 * g + lk] : x[j][a][g];
#pragma unroll
                    for (int b = 0; b < 10; ++b) dacc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[j][BJ[b]][g], xs[BI[b]], dacc[b], 0, 0, 0);
                }
            }
        // the shares: slot 3 b + (rank of the writer among the three waves that do not own block b), 256 doubles each, [q][lane]
        double *Ps = scratch;
#pragma unroll
        for (int b = 0; b < 10; ++b)
            if (wave != OW[b]) {
                double *slot = Ps + (3 * b + (wave < OW[b] ? wave : wave - 1)) * 256 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) slot[64 * q] = dacc[b][q];
            }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < nblk) {
                // the wave's own share of its t-th block
                v4f64 own = wave == 0 ? dacc[t] : (wave == 1 ? dacc[3 + t] : (wave == 2 ? dacc[t < 2 ? 6 + t : 7] : dacc[t < 2 ? 8 + t : 9]));
                const double *sl = Ps + 3 * (bfirst + t) * 256 + lane;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    v4f64 part;
                    if (w == wave) part = own;
                    else {
                        const double *sw = sl + (w < wave ? w : w - 1) * 256;
#pragma unroll
                        for (int q = 0; q < 4; ++q) part[q] = sw[64 * q];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) tot[t][q] = (w == 0) ? part[q] : tot[t][q] + part[q];
                }
            }
        __syncthreads();                                         // the shares are read: the image (same LDS) is next
    } else {
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < nblk) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pvt[t][q] = P[(i64)min(16 * bi[t] + lr, nb - 1) + (i64)min(16 * bj[t] + lk + 4 * q, nb - 1) * lda];
            }
    }
    if (SIGNED && tid < NB_IN) Sg[tid] = (tid < nb) ? sg[bk0 + tid] : 1.0;
    // As = block - sum: the blocks below the diagonal blocks + FULL (symmetric) diagonal 16 x 16 blocks; rows / columns beyond nb: identity
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < nblk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 16 * bi[t] + lr, col = 16 * bj[t] + lk + 4 * q;
                const double v = (row < nb && col < nb) ? (pvt[t][q] - tot[t][q]) : ((row == col) ? 1.0 : 0.0);
                if (row >= col) {
                    Mt[col * PD_LD + row] = v;
                    if (row > col && bi[t] == bj[t]) Mt[row * PD_LD + col] = v;
                }
            }
        }
#else
    double pv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) pv[q] = P[(i64)min(r, nb - 1) + (i64)min(wave + 4 * q, nb - 1) * lda];
    // left-looking over the already factored 64-wide steps of this block column.  Round 6: the rows of this block in a solved step -- a 64 x 64 block of L, both
    // operands of D -= X S X' -- travel global -> registers -> LDS ONCE (16 loads per thread and step, all steps' loads issued together with the block's own
    // entries) instead of 20 operand loads per wave and 16 columns straight from the panel (a round trip per 16 columns: 12.5 us in front of the fourth block of
    // a block column, for 5 us of matrix-core time); and the ten 16 x 16 blocks on and below the diagonal are dealt 3 + 3 + 2 + 2 to the waves (by rows a wave
    // formed four blocks, of which wave 0 needed one).  Every entry sums its columns in ascending order, four per v_mfma_f64_16x16x4_f64: same bits as before.
    // The shared block lies BEHIND the image (Xs[r * TRM_LDT + k], the layout of trsm_rows_mt's Wt), Ds in the image's place as before.
    const i32 Kp = bk0 - kprev;
    if (Kp > 0) {
        double *Xs = scratch + NB_IN * PD_LD;
        const int nprev = Kp / NB_IN;                            // 1 .. 3 (workgroup-uniform)
        double st[3][16];
        const i32 rsrc = bk0 + min(lane, nb - 1);                // row of the block this thread carries (clamped: the rows >= nb feed entries nobody reads)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nprev) {
#pragma unroll
                for (int i = 0; i < 16; ++i) st[j][i] = pcol(c, fd, kprev + j * NB_IN + wave + 4 * i)[rsrc];
            }
        // blocks (bi, bj) of this wave: wave 0: (0,0) (3,0) (3,1) | 1: (1,0) (1,1) (3,2) | 2: (2,0) (2,1) | 3: (2,2) (3,3)
        const int nblk = wave < 2 ? 3 : 2;
        int bi[3], bj[3];
        bi[0] = wave == 3 ? 2 : wave; bj[0] = wave == 3 ? 2 : 0;
        bi[1] = wave < 2 ? (wave == 0 ? 3 : 1) : (wave == 2 ? 2 : 3); bj[1] = wave == 0 ? 0 : (wave == 3 ? 3 : 1);
        bi[2] = 3; bj[2] = wave == 0 ? 1 : 2;
        v4f64 dacc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) dacc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nprev) {
                if (j > 0) __syncthreads();
#pragma unroll
                for (int i = 0; i < 16; ++i) Xs[lane * TRM_LDT + wave + 4 * i] = st[j][i];
                __syncthreads();
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if (t < nblk) {
                        // (both operands of the block's 16 products to registers first: the reads of the next block are in flight behind this block's products)
                        const double *Xb = Xs + (16 * bi[t] + lr) * TRM_LDT + lk, *Xa = Xs + (16 * bj[t] + lr) * TRM_LDT + lk;
                        double xa[16], xb[16];
#pragma unroll
                        for (int ks = 0; ks < 16; ++ks) {
                            xa[ks] = Xa[4 * ks];
                            xb[ks] = Xb[4 * ks];
                            if (SIGNED) xb[ks] *= sg[kprev + j * NB_IN + 4 * ks + lk];
                        }
#pragma unroll
                        for (int ks = 0; ks < 16; ++ks) dacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks], xb[ks], dacc[t], 0, 0, 0);
                    }
            }
        __syncthreads();                                         // the last block is read: Ds (the image's place) is next, then Sg (inside Xs)
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < nblk) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Ds[(16 * bj[t] + lk + 4 * q) * PD_LD + 16 * bi[t] + lr] = dacc[t][q];
            }
    }
    if (SIGNED && tid < NB_IN) Sg[tid] = (tid < nb) ? sg[bk0 + tid] : 1.0;
    if (Kp > 0) __syncthreads();
    // As = block - Ds in place of Ds: the blocks below the diagonal blocks + FULL (symmetric) diagonal 16 x 16 blocks; rows / columns beyond nb: identity
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int col = wave + 4 * q;
        const double dv = (Kp > 0) ? Ds[col * PD_LD + r] : 0.0;
        const double v = (rok && col < nb) ? (pv[q] - dv) : ((r == col) ? 1.0 : 0.0);
        if (r >= col) {
            Mt[col * PD_LD + r] = v;
            if (r > col && (r >> 4) == (col >> 4)) Mt[r * PD_LD + col] = v;
        }
    }
#endif
    // (prog, k_chain with early strips: everything this workgroup stored so far -- the steps in front of this block, factored and solved inside the block column --
    //  has left the CU before the barrier; wave 1, idle beside wave 0's first panel, then publishes it: one release, the block column's counter + 1)
    if (prog) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PT_STAMP();
    i32 failcol = NB_IN;                                         // first pivot of the wrong sign (wave 0; wave-uniform)
    // one 16 x 16 tile of the part not yet factored: T(g, q) -= L_gp S L_qp'
    auto tile_update = [&](const int g, const int q, const int p) {
        v4f64 acc;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) acc[qq] = Mt[(PW_W * q + lk + 4 * qq) * PD_LD + PW_W * g + lr];
#pragma unroll
        for (int k4 = 0; k4 < PW_W; k4 += 4) {
            const int kk = PW_W * p + k4 + lk;
            double x = -Mt[kk * PD_LD + PW_W * q + lr];
            if (SIGNED) x *= Sg[kk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, Mt[kk * PD_LD + PW_W * g + lr], acc, 0, 0, 0);
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) Mt[(PW_W * q + lk + 4 * qq) * PD_LD + PW_W * g + lr] = acc[qq];
    };
    // columns [c0, c1) of the finished panel pp to the packed panel in global memory (lane = row: one 512-byte segment per column)
    auto store_panel = [&](const int pp, const int c0, const int c1) {
        const double *Lq = Lp + (pp & 1) * PD_LP;
        int r = lane;
        asm volatile("" : "+v"(r));
        const bool rok = r < nb;
        const int rd = min(max(r - PW_W * pp, 0), PW_W - 1);
#pragma unroll 4
        for (int cc = c0; cc < c1; ++cc) {
            const i32 col = PW_W * pp + cc;
            const double vb = Mt[col * PD_LD + r], vd = Lq[cc * PW_W + rd];
            if (r >= col && rok && col < nb) P[(i64)r + (i64)col * lda] = (r < PW_W * (pp + 1)) ? vd : vb;
        }
    };
    // inverse of the diagonal block pp (one wave; every row of 16 lanes computes it, row g stores its share)
    auto inv_diag = [&](const int pp) {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int g = lo >> 4, cc = lo & 15;
        double l[PW_W], x[PW_W];
#pragma unroll
        for (int k = 0; k < PW_W; k += 2) {
            const v2f64 t = *reinterpret_cast<const v2f64 *>(Lp + (pp & 1) * PD_LP + cc * PW_W + k);
            l[k] = t.x; l[k + 1] = t.y;
        }
        double lcc = 1.0;
#pragma unroll
        for (int k = 0; k < PW_W; ++k) { lcc = (k == cc) ? l[k] : lcc; x[k] = (k == cc) ? 1.0 : 0.0; }
        const double il = 1.0 / lcc;
        dpp_inv_steps<0>(l, x, il);
        const i32 col = PW_W * pp + cc;
        if (g == pp) {
#pragma unroll
            for (int k = 0; k < PW_W; ++k) {
                Mt[(PW_W * pp + k) * PD_LD + col] = x[k];                          // W row-major in the diagonal block
                Wd[pp * PW_W * PW_LDT + cc * PW_LDT + k] = x[k];
            }
        }
        if (g <= pp) {                                                            // rows of block g of column `col`: zeros above the diagonal block
            if (wfull) {
#pragma unroll
                for (int k = 0; k < PW_W; k += 2)
                    *reinterpret_cast<v2f64 *>(W + (i64)col * NB_IN + PW_W * g + k) = (g == pp) ? (v2f64){x[k], x[k + 1]} : (v2f64){0.0, 0.0};
            } else if (col < nb) {
#pragma unroll
                for (int k = 0; k < PW_W; ++k)
                    if (PW_W * g + k < nb) W[(i64)col * nb + PW_W * g + k] = (g == pp) ? x[k] : 0.0;
            }
        }
    };
    // Off-diagonal blocks of the inverse: W_ij = -W_ii G_ij, G_ij = sum_{k=j}^{i-1} L_ik W_kj (needs the blocks W_kj of the row blocks above i).
    // inv_gpart adds the terms k0 <= k < k1 to an accumulator (matrix-core layout), inv_finish multiplies by -W_ii through the wave's transposition piece.
    auto inv_gpart = [&](v4f64 gq, const int i, const int j, const int k0, const int k1) {
#pragma unroll 1
        for (int k = k0; k < k1; ++k) {
#pragma unroll
            for (int k4 = 0; k4 < PW_W; k4 += 4)
                gq = __builtin_amdgcn_mfma_f64_16x16x4f64(Mt[(PW_W * k + k4 + lk) * PD_LD + PW_W * j + lr], Mt[(PW_W * k + k4 + lk) * PD_LD + PW_W * i + lr], gq, 0, 0, 0);
        }
        return gq;
    };
    auto inv_finish = [&](const v4f64 gq, const int i, const int j) {
        double *Tw = Ts + wave * (PW_W * PW_LDT);
#pragma unroll
        for (int q = 0; q < 4; ++q) Tw[lr * PW_LDT + lk + 4 * q] = gq[q];
        TLPK_LDS_FENCE();
        v4f64 h = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k4 = 0; k4 < PW_W; k4 += 4)
            h = __builtin_amdgcn_mfma_f64_16x16x4f64(Tw[(k4 + lk) * PW_LDT + lr], Wd[i * PW_W * PW_LDT + (k4 + lk) * PW_LDT + lr], h, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = PW_W * i + lr, cq = PW_W * j + lk + 4 * q;
            Mt[rr * PD_LD + cq] = -h[q];
            if (rr < nb && cq < nb) W[(i64)rr + (i64)cq * nb] = -h[q];
        }
        TLPK_LDS_FENCE();
    };
    const v4f64 zero4 = {0.0, 0.0, 0.0, 0.0};
    v4f64 ga = zero4, gb = zero4;                                // sums G_ij of waves 0, 2, 3, kept in registers from the slot that forms them to the phase that finishes them
    // What is left of panel pp (pp < 3) once the next panel has what it reads -- beside wave 0's panel pp + 1:
    //   wave 1: the inverse W_pp of the diagonal block (a serial chain of its own: ~1 us);
    //   waves 2, 3: the tiles panel pp + 1 does not read, the stores of panel pp, and every sum G_ij whose terms are older than the last barrier.
    // The products -W_ii G_ij follow in the next phase in which W_ii is visible and a wave is idle (W_10: wave 3 beside the tile of panel 3;
    // W_20, W_21: waves 2, 3 beside W_33; W_3j: after one more barrier).
    auto lazy = [&](const int pp) {
        if (wave == 1) inv_diag(pp);
        else {
            int idx = 0;
            for (int q = pp + 2; q < 4; ++q)
                for (int g = q; g < 4; ++g, ++idx)
                    if ((idx & 1) == (wave & 1)) tile_update(g, q, pp);
            store_panel(pp, wave == 2 ? 0 : PW_W / 2, wave == 2 ? PW_W / 2 : PW_W);
            if (pp == 1 && wave == 3) ga = inv_gpart(zero4, 1, 0, 0, 1);           // G_10 = L_10 W_00
            if (pp == 2) {
                ga = inv_gpart(zero4, 2, wave - 2, wave - 2, 2);                   // G_20 = L_20 W_00 + L_21 W_10 (wave 2), G_21 = L_21 W_11 (wave 3)
                gb = inv_gpart(zero4, 3, wave - 2, wave - 2, 2);                   // the terms k < 2 of G_30 (wave 2), G_31 (wave 3)
            }
        }
    };
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
        if (wave == 0) {
            switch (p) {
            case 0: dpp_panel<SIGNED, 0>(Mt, Lp, Sg, lane, failcol); break;
            case 1: dpp_panel<SIGNED, 1>(Mt, Lp, Sg, lane, failcol); break;
            case 2: dpp_panel<SIGNED, 2>(Mt, Lp, Sg, lane, failcol); break;
            default: dpp_panel<SIGNED, 3>(Mt, Lp, Sg, lane, failcol); break;
            }
        } else if (p > 0) lazy(p - 1);                           // beside wave 0's panel p
        else if (prog && wave == 1) {
            if (lane == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(prog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        PT_STAMP();
        __syncthreads();                                         // panel p of L is in Mt / Lp; the lazy work of panel p - 1 is done
        PT_STAMP();
        if (p < 3) {
            if (wave >= 1 && p + wave <= 3) tile_update(p + wave, p + 1, p);      // what panel p + 1 reads
            if (p == 2 && wave == 3) inv_finish(ga, 1, 0);                        // W_10 = -W_11 G_10 (W_11: beside panel 2)
            __syncthreads();
            PT_STAMP();
        }
    }
    // the end: W_33 by wave 1; beside it waves 2, 3 finish W_20, W_21 (W_22: beside panel 3) and add their term to G_30, G_31, wave 0 forms G_32 and
    // stores panel 3; after one more barrier W_3j = -W_33 G_3j
    if (wave == 1) inv_diag(3);
    else if (wave == 0) {
        if (failcol < NB_IN && lane == 0) atomicMin(c.info, fd.col0 + bk0 + failcol);
        gb = inv_gpart(zero4, 3, 2, 2, 3);
        store_panel(3, 0, PW_W);
    } else {
        inv_finish(ga, 2, wave - 2);
        gb = inv_gpart(gb, 3, wave - 2, 2, 3);
    }
    PT_STAMP();
    __syncthreads();
    if (wave != 1) inv_finish(gb, 3, wave == 0 ? 2 : wave - 2);
    PT_STAMP();
}

// ------------------------------------------------------------------------------------------
// trsm on the matrix cores: X = B * L11^{-T} as the product with the inverted diagonal block,
// computed transposed (D[c][r] = sum_k Linv[c][k] * B[r][k]) so that consecutive lanes write
// consecutive panel rows.  The operand fragments of a 16-row strip stay in registers: each
// element is needed by exactly one lane, and the accumulator layout of D coincides lane by lane
// with the operand layout (element (a, q) <-> fragment 4a + q), so a solved 64-column step feeds
// the following steps without leaving the registers.  The 64 x 64 blocks of L / Linv are shared
// through LDS.
// ------------------------------------------------------------------------------------------
constexpr int LDW = NB_IN + 16;                 // == 16 mod 32: conflict-free ds_read_b64
static_assert(NB_IN * LDW >= POTRF_SCRATCH && NB_IN * LDW >= POTRF_PAIR_SCRATCH && POTRF_DPP_LDS >= POTRF_WAVE_LDS, "the trsm LDS block doubles as the potrf scratch");

// Pivot block of a small front (ns <= SMALL_NS), one WAVE per front, four fronts per workgroup: lane i
// holds row i of the lower triangle and row i of an identity block in registers, pivots and pivot
// columns travel by shuffles.  Same algorithm as potrf_block: unit-lower eliminations applied to
// [A | I], columns scaled by 1/sqrt(d), inverse rows scaled by 1/L_ii; a non-positive pivot records
// its column and is replaced by 1.
template <bool SIGNED>
__global__ __launch_bounds__(256) void k_potrf_small(const PotrfTask *__restrict__ tasks, DevCtx c) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform: task and front descriptor in scalar registers)
    const PotrfTask t = tasks[(i64)blockIdx.x * 4 + wave];
    if (t.front < 0) return;
    const FrontDesc fd = c.fronts[t.front];
    const double *sg = SIGNED ? c.csign + fd.col0 : nullptr;
    const i32 ns = fd.ns;
    const i32 lda = fd.lda;                         // leading dimension of the panel (>= f, multiple of 16 for large fronts)
    double *P = c.Lval + fd.loff;
    const i32 ic = min(lane, ns - 1);
    double a[SMALL_NS], w[SMALL_NS];
#pragma unroll
    for (int j = 0; j < SMALL_NS; ++j) {
        const double v = P[(i64)ic + (i64)min(j, ns - 1) * lda];           // clamped, unconditional
        a[j] = (lane < ns && j < ns && j <= lane) ? v : 0.0;
        w[j] = (j == lane) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int k = 0; k < SMALL_NS; ++k) {
        if (k < ns) {                                                   // wave-uniform
            double d = __shfl(a[k], k);
            const double sk = SIGNED ? sg[k] : 1.0;
            if (!(sk * d > 0.0)) {
                if (lane == 0) atomicMin(c.info, fd.col0 + k);
                d = sk;
            }
            if (SIGNED) d = fabs(d);
            double isq = __builtin_amdgcn_rsq(d);
            isq = isq * (1.5 - 0.5 * d * isq * isq);
            isq = isq * (1.5 - 0.5 * d * isq * isq);
            double sq = d * isq;
            sq = fma(0.5 * isq, fma(-sq, sq, d), sq);
            const double m = (lane > k) ? a[k] * (isq * isq) * sk : 0.0; // multiplier a_ik / d
            if (SIGNED) isq *= sk;
#pragma unroll
            for (int j = k + 1; j < SMALL_NS; ++j) {
                const double ajk = __shfl(a[k], j);                     // a_jk from lane j
                a[j] -= (j <= lane) ? m * ajk : 0.0;
            }
#pragma unroll
            for (int cc = 0; cc <= k; ++cc) {
                const double wkc = __shfl(w[cc], k);                    // row k of the identity block
                w[cc] -= m * wkc;
            }
            a[k] = (lane == k) ? sq : ((lane > k) ? a[k] * isq : a[k]);
        }
    }
    // lane i: a[i] = L_ii.  Dynamic index -> select chain.
    double lii = 1.0;
#pragma unroll
    for (int j = 0; j < SMALL_NS; ++j) lii = (j == lane) ? a[j] : lii;
    const double ili = 1.0 / lii;
    double *W = front_dinv(c, fd, 0);                                    // ns x ns, column-major, ld = ns
    if (lane < ns) {
#pragma unroll
        for (int j = 0; j < SMALL_NS; ++j) {
            if (j < ns) {
                if (j <= lane) P[(i64)lane + (i64)j * lda] = a[j];
                W[(i64)lane + (i64)j * ns] = (j <= lane) ? w[j] * ili : 0.0;
            }
        }
    }
}

// Rows [row0, min(row0 + 64 NBR, rowlim)) of the 64-wide step k0 (width nb), inside the diagonal
// block of a block column: B -= X_prev * L[k0.., kprev..k0)' first (left-looking), then the
// solve.  One wave = 16 rows of each 64-row block (wave w: rows row0 + 16 w + 64 b, b < NBR).  In place: a wave only overwrites its own rows, after
// all of its loads.  NBR = 3 (round 5): ALL the rows below the step inside a 256-wide block column in one call -- the staged blocks (L of the
// earlier steps, the inverted diagonal block) are shared by the row blocks, a block column costs 3 calls instead of 6 (every row is computed
// exactly as by the one-block calls: same sums, same order).
template <bool SIGNED = false, int NBR = 1>
__device__ __forceinline__ void trsm_rows(const DevCtx &c, const FrontDesc &fd, const i32 k0, const i32 nb,
                                          const i32 row0, const i32 rowlim, const i32 kprev, double *Ws) {
    const double *sg = SIGNED ? c.csign + fd.col0 : nullptr;
    // packed panel: the step's columns [k0, k0 + nb) lie in one 64-column slice (k0, kprev multiples of 64), the solved steps
    // [kprev + c0, kprev + c0 + 64) in one each
    double *P0 = pcol(c, fd, k0);                   // virtual row 0 of column k0
    const i32 ld0 = pld(fd, k0);
    const double *W = front_dinv(c, fd, k0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const i32 rbase = row0 + wave * 16;
    const int nact = (rbase < rowlim) ? min(NBR, (rowlim - rbase + NB_IN - 1) / NB_IN) : 0;      // (wave-uniform) 16-row groups of this wave
    // rows are clamped instead of guarded (per-lane guards turn every load into its own
    // exec-masked branch with its own wait; a clamped row only produces entries the stores skip)
    double bf[NBR][16];
    i32 rowc[NBR];
#pragma unroll
    for (int b = 0; b < NBR; ++b) rowc[b] = min(rbase + b * NB_IN + lr, rowlim - 1);
#pragma unroll
    for (int b = 0; b < NBR; ++b)
        if (b < nact) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const i32 k = 4 * ks + lk;
                const double v = P0[(i64)rowc[b] + (i64)min(k, nb - 1) * ld0];      // clamped, not guarded
                bf[b][ks] = (k < nb) ? v : 0.0;
            }
        }
    v4f64 acc[4];
    const i32 Kp = k0 - kprev;
    for (i32 c0 = 0; c0 < Kp; c0 += NB_IN) {
        __syncthreads();
        const double *Pc = pcol(c, fd, kprev + c0);
        const i32 ldc = pld(fd, kprev + c0);
        for (int idx = tid; idx < NB_IN * NB_IN; idx += 256) {
            const int cc = idx & (NB_IN - 1), k = idx >> 6;
            const double v = Pc[(i64)(k0 + min(cc, nb - 1)) + (i64)k * ldc];
            Ws[k * LDW + cc] = (cc < nb) ? v : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NBR; ++b)
            if (b < nact) {
                double xf[16];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    xf[ks] = Pc[(i64)rowc[b] + (i64)(4 * ks + lk) * ldc] * (SIGNED ? sg[kprev + c0 + 4 * ks + lk] : 1.0);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[(4 * ks + lk) * LDW + a * 16 + lr], xf[ks], acc[a], 0, 0, 0);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bf[b][4 * a + q] -= acc[a][q];
            }
    }
    __syncthreads();
    for (int idx = tid; idx < NB_IN * NB_IN; idx += 256) {
        const int cc = idx & (NB_IN - 1), k = idx >> 6;
        const double v = W[(i64)min(cc, nb - 1) + (i64)min(k, nb - 1) * nb];
        Ws[k * LDW + cc] = (cc < nb && k < nb) ? v : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NBR; ++b)
        if (b < nact) {
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (4 * ks > 16 * a + 15) continue;       // Linv[c][k] = 0 for k > c: whole block is zero
                    acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[(4 * ks + lk) * LDW + a * 16 + lr], bf[b][ks], acc[a], 0, 0, 0);
                }
            }
            // D[i][j] (reg q: i = lk + 4q -> column c = 16a + i ; j = lr -> row)
            const i32 row = rbase + b * NB_IN + lr;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const i32 cc = a * 16 + lk + 4 * q;
                    if (row < rowlim && cc < nb) P0[(i64)row + (i64)cc * ld0] = SIGNED ? acc[a][q] * sg[k0 + cc] : acc[a][q];
                }
        }
}

// Round 6: trsm_rows behind potrf_block_dpp, step SI (0, 1, 2) of a block column -- the rows [row0, rowlim) below the 64 x 64 block that potrf_block_dpp has
// just factored (all 64 columns: the caller only comes here when another step follows), SI solved steps in front of it, at most 3 - SI row blocks.
// The arithmetic is trsm_rows's, entry by entry and in the same order (same sums, same bits); what changes is where the operands come from and when:
//   * EVERY global load of the call is issued before the first product: the rows' own entries, their solved steps (xf), and the 64 x 64 blocks of L the
//     left-looking products share (16 entries per thread and block, kept in registers until their turn in LDS) -- one round trip per call instead of one
//     per staged block and one per operand (five for step 2; the block column's in-block solves were 43 of its 115 us for 13 us of matrix-core time);
//   * the inverted diagonal block is read where potrf_block_dpp left it: Mt[r * PD_LD + c] = W[r][c] for every 16 x 16 block on or below the diagonal
//     (the blocks above it are never multiplied).  Lane stride PD_LD = 82 doubles == 18 mod 32: the 16 x 2 lanes of a half wave hit 32 different bank pairs.
//   * the shared blocks of L go through Wt, BEHIND Mt (Wt[cc * TRM_LDT + k] = L[k0 + cc][kprev + 64 j + k]: the transposed layout, read like Mt).
template <bool SIGNED, int SI>
__device__ __forceinline__ void trsm_rows_mt(const DevCtx &c, const FrontDesc &fd, const i32 k0, const i32 row0, const i32 rowlim, const i32 kprev,
                                             const double *Mt, double *Wt) {
    constexpr int NBR = 3 - SI, NPREV = SI;
    const double *sg = SIGNED ? c.csign + fd.col0 : nullptr;
    double *P0 = pcol(c, fd, k0);
    const i32 ld0 = pld(fd, k0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const i32 rbase = row0 + wave * 16;
    const int nact = (rbase < rowlim) ? min(NBR, (rowlim - rbase + NB_IN - 1) / NB_IN) : 0;      // (wave-uniform) 16-row groups of this wave
    double bf[NBR][16], xf[NPREV > 0 ? NPREV : 1][NBR][16], st[NPREV > 0 ? NPREV : 1][16];
    i32 rowc[NBR];
#pragma unroll
    for (int b = 0; b < NBR; ++b) rowc[b] = min(rbase + b * NB_IN + lr, rowlim - 1);              // clamped, not guarded (trsm_rows)
#pragma unroll
    for (int b = 0; b < NBR; ++b)
        if (b < nact) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) bf[b][ks] = P0[(i64)rowc[b] + (i64)(4 * ks + lk) * ld0];
        }
#pragma unroll
    for (int j = 0; j < NPREV; ++j) {
        const double *Pc = pcol(c, fd, kprev + j * NB_IN);
        const i32 ldc = pld(fd, kprev + j * NB_IN);
#pragma unroll
        for (int i = 0; i < 16; ++i) st[j][i] = Pc[(i64)(k0 + lane) + (i64)(wave + 4 * i) * ldc];         // L[k0 + cc][.. + k], cc = lane, k = wave + 4 i
#pragma unroll
        for (int b = 0; b < NBR; ++b)
            if (b < nact) {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    xf[j][b][ks] = Pc[(i64)rowc[b] + (i64)(4 * ks + lk) * ldc] * (SIGNED ? sg[kprev + j * NB_IN + 4 * ks + lk] : 1.0);
            }
    }
    // one LDS read of a shared operand serves every row block of the wave (ks, a outside, b inside: an entry still sums its columns in ascending order)
    v4f64 acc[NBR][4];
#pragma unroll
    for (int j = 0; j < NPREV; ++j) {
        if (j > 0) __syncthreads();                              // the products of block j - 1 are done with Wt
#pragma unroll
        for (int i = 0; i < 16; ++i) Wt[lane * TRM_LDT + wave + 4 * i] = st[j][i];
        __syncthreads();
        if (nact > 0) {
#pragma unroll
            for (int b = 0; b < NBR; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[b][a] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const double wv = Wt[(a * 16 + lr) * TRM_LDT + 4 * ks + lk];
#pragma unroll
                    for (int b = 0; b < NBR; ++b)
                        if (b < nact) acc[b][a] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, xf[j][b][ks], acc[b][a], 0, 0, 0);
                }
            }
#pragma unroll
            for (int b = 0; b < NBR; ++b)
                if (b < nact) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int q = 0; q < 4; ++q) bf[b][4 * a + q] -= acc[b][a][q];
                }
        }
    }
    if (nact > 0) {
#pragma unroll
        for (int b = 0; b < NBR; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[b][a] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (4 * ks > 16 * a + 15) continue;           // Linv[c][k] = 0 for k > c: whole block is zero
                const double wv = Mt[(a * 16 + lr) * PD_LD + 4 * ks + lk];
#pragma unroll
                for (int b = 0; b < NBR; ++b)
                    if (b < nact) acc[b][a] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, bf[b][ks], acc[b][a], 0, 0, 0);
            }
        }
#pragma unroll
        for (int b = 0; b < NBR; ++b)
            if (b < nact) {
                const i32 row = rbase + b * NB_IN + lr;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const i32 cc = a * 16 + lk + 4 * q;
                        if (row < rowlim) P0[(i64)row + (i64)cc * ld0] = SIGNED ? acc[b][a][q] * sg[k0 + cc] : acc[b][a][q];
                    }
            }
    }
}

// Diagonal block of one block column (t.nb <= NB_OUT columns from t.k0; the columns before k0 have
// already been applied by the left-looking k_update): 64-wide steps, each a potrf of the step's
// diagonal block followed by the trsm of the rows below it INSIDE the block -- one workgroup runs
// the whole chain, so the factorisation's critical path costs one launch per block column.
// WAVE: the 64 x 64 diagonal blocks by potrf_block_wave (one wave, no barrier per column) instead of potrf_block (TLPK_POTRF_WAVE=0: the
// round-1..3 kernels)
static_assert(LDW_ == LDW, "potrf_block_wave: LDS stride");
// MODE: 0 = potrf_block (one barrier per column), 1 = potrf_block_wave, 2 = potrf_block_pair (one barrier per two columns), 3 = potrf_block_dpp (round 5; with ONE
// trsm_rows call per 64-wide step inside k_potrf_wide: 115 against 123 us per 256-wide block, same bits)
template <bool SIGNED, int MODE>
__device__ __forceinline__ void potrf_block_any(const DevCtx &c, const FrontDesc &fd, const i32 bk0, const i32 nb, const i32 kprev, double *scratch) {
    if (MODE == 3) potrf_block_dpp<SIGNED>(c, fd, bk0, nb, kprev, scratch);
    else if (MODE == 1) potrf_block_wave<SIGNED>(c, fd, bk0, nb, kprev, scratch);
    else if (MODE == 2) potrf_block_pair<SIGNED>(c, fd, bk0, nb, kprev, scratch);
    else potrf_block<SIGNED>(c, fd, bk0, nb, kprev, scratch);
}
template <bool SIGNED, int MODE>
__global__ __launch_bounds__(256) void k_potrf(const PotrfTask *__restrict__ tasks, DevCtx c) {      // t.nb <= NB_IN
    __shared__ __attribute__((aligned(16))) double scratch[MODE == 3 ? POTRF_DPP_LDS : (MODE == 1 ? POTRF_WAVE_LDS : (MODE == 2 ? POTRF_PAIR_SCRATCH : POTRF_SCRATCH))];
    const PotrfTask t = tasks[blockIdx.x];
    const FrontDesc fd = c.fronts[t.front];
    potrf_block_any<SIGNED, MODE>(c, fd, t.k0, t.nb, t.k0, scratch);
}
template <bool SIGNED, int MODE>
__device__ __forceinline__ void potrf_wide_task(const PotrfTask t, const FrontDesc &fd, const DevCtx &c, double *Ws, unsigned *prog = nullptr) {
    const i32 k0 = t.k0, w = t.nb, kend = k0 + w;
    // the chain's four waves are served before the update tiles' waves that share their SIMDs (beside k_update the chain runs on the side stream: its end is what the
    // group's stream waits for at the join): C4 51.0 -> 50.8 ms, pds-class 15.79 -> 15.68 ms (profiles/r05_chain_overlap.txt)
    if (MODE == 3) __builtin_amdgcn_s_setprio(3);
    potrf_block_any<SIGNED, MODE>(c, fd, k0, min(w, NB_IN), k0, Ws);
    if constexpr (MODE == 3 && TRM_MT) {                         // round 6: the in-block solves read W from potrf_block_dpp's LDS image, one round of global loads per step
        double *Wt = Ws + NB_IN * PD_LD;
#ifdef POTRF_TRACE   /* tools/potrf_wave_bench.hip: phase boundaries of the block column in words 16.. of the workgroup's row */
        unsigned long long *ptw_ = (unsigned long long *)c.spart + (size_t)blockIdx.x * 32 + 16;
#define PTW_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0) ptw_[i] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PTW_STAMP(i) do {} while (0)
#endif
        PTW_STAMP(0);
        if (k0 + NB_IN < kend) {
            __syncthreads();                                     // own global stores visible, the image complete, the LDS behind it free
            PTW_STAMP(1);
            trsm_rows_mt<SIGNED, 0>(c, fd, k0, k0 + NB_IN, kend, k0, Ws, Wt);
            __syncthreads();
            PTW_STAMP(2);
            potrf_block_dpp<SIGNED>(c, fd, k0 + NB_IN, min(NB_IN, kend - (k0 + NB_IN)), k0, Ws, prog);
        }
        if (k0 + 2 * NB_IN < kend) {
            __syncthreads();
            PTW_STAMP(3);
            trsm_rows_mt<SIGNED, 1>(c, fd, k0 + NB_IN, k0 + 2 * NB_IN, kend, k0, Ws, Wt);
            __syncthreads();
            PTW_STAMP(4);
            potrf_block_dpp<SIGNED>(c, fd, k0 + 2 * NB_IN, min(NB_IN, kend - (k0 + 2 * NB_IN)), k0, Ws, prog);
        }
        if (k0 + 3 * NB_IN < kend) {
            __syncthreads();
            PTW_STAMP(5);
            trsm_rows_mt<SIGNED, 2>(c, fd, k0 + 2 * NB_IN, k0 + 3 * NB_IN, kend, k0, Ws, Wt);
            __syncthreads();
            PTW_STAMP(6);
            potrf_block_dpp<SIGNED>(c, fd, k0 + 3 * NB_IN, min(NB_IN, kend - (k0 + 3 * NB_IN)), k0, Ws, prog);
        }
        PTW_STAMP(7);
#undef PTW_STAMP
        return;
    }
    for (i32 ks = k0; ks + NB_IN < kend; ks += NB_IN) {          // step ks is factored: rows below, next diagonal block
        if constexpr (MODE == 3) {
            __syncthreads();                                     // own global stores visible, Ws free
            trsm_rows<SIGNED, (NB_OUT - NB_IN) / NB_IN>(c, fd, ks, NB_IN, ks + NB_IN, kend, k0, Ws);      // all the rows below the step inside the block column
        } else {                                                 // (the older block kernels keep their one-block calls: next to them three row blocks in registers spill)
            for (i32 r0 = ks + NB_IN; r0 < kend; r0 += NB_IN) {
                __syncthreads();
                trsm_rows<SIGNED>(c, fd, ks, NB_IN, r0, kend, k0, Ws);
            }
        }
        __syncthreads();
        potrf_block_any<SIGNED, MODE>(c, fd, ks + NB_IN, min(NB_IN, kend - (ks + NB_IN)), k0, Ws);
    }
}
template <bool SIGNED, int MODE>
__global__ __launch_bounds__(256, 2) void k_potrf_wide(const PotrfTask *__restrict__ tasks, DevCtx c) {
    __shared__ __attribute__((aligned(16))) double Ws[MODE == 3 ? POTRF_WIDE_DPP_LDS : (MODE == 1 ? POTRF_WAVE_LDS : NB_IN * LDW)];
    const PotrfTask t = tasks[blockIdx.x];
    const FrontDesc fd = c.fronts[t.front];
    potrf_wide_task<SIGNED, MODE>(t, fd, c, Ws);
}

// Rows below the diagonal block of a block column: X = B * L11^{-T} for the whole (<= 256 wide)
// block column in ONE pass: a wave keeps its 16 rows x 256 columns in registers, and walks the
// 64-wide steps: B_i -= sum_{j<i} X_j L_ij' ; X_i = B_i Linv_ii'.  The panel is read once and
// written once (the stepwise variant re-read the solved steps of the block column from HBM for
// every following step).
// SIGNED: the registers keep B Linv' = X S (what the later steps of the block column need: B_i -= sum_j X_j S_j L_ij');
// the stored factor block X gets its column signs only at the final store.
template <bool SIGNED, bool FULLW>
__device__ __forceinline__ void trsm_task(const TrsmTask t, const FrontDesc &fd, const DevCtx &c, double (*Wb)[NB_IN * LDW]) {
    // two operand buffers: the block staged for step n + 1 never overwrites what slower waves still read for step n, so a
    // step needs ONE barrier (after its stores) instead of two
    int wsel = 0;
    const i32 k0 = t.k0, w = t.nb;
    const i32 lda = fd.lda;                         // leading dimension of the panel (>= f, multiple of 16 for large fronts)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const i32 rbase = t.row0 + wave * 16;
    const i32 rlim = t.pad1;                        // rows [row0, rlim) belong to this task
    const bool active = rbase < rlim;
    const i32 rowc = min(rbase + lr, rlim - 1);     // clamped, stores are guarded
    double bf[4][16];                               // bf[i][ks] = B[row][k0 + 64 i + 4 ks + lk]
    // packed panel: every 64-column step of the block column is one slice (k0 is a multiple of 64); columns beyond w read the
    // step's last column (in bounds) and count as zero.  The rows of step i are loaded two steps ahead of their use (not all
    // four up front: the registers of the late steps are free for the operand prefetch meanwhile, nothing is spilled while
    // the loads are in flight).
    // (round 6) Full-width block columns (w == 256, all but a front's last): addresses as a wave-uniform base -- the step's slice, advanced by four columns per
    // fragment in scalar registers -- plus ONE 32-bit lane offset (row + lk columns), for the loads of the rows, the fetches of the shared blocks and the stores.
    // The general form computes a 64-bit address per lane and load (clamped columns): in a kernel that holds 16 rows x 256 columns per wave in registers that was
    // 40 spilled registers and 164 bytes of scratch per lane (92 scratch instructions between the products).
    constexpr bool fullw = FULLW;                                     // (the caller checks w == NB_OUT)
    auto load_step = [&](const int i) {
        if constexpr (fullw) {
            const i32 ldi = lda - (k0 + 64 * i);
            const char *Pb = reinterpret_cast<const char *>(pcol(c, fd, k0 + 64 * i));      // (uniform)
            unsigned vo = (unsigned)((i64)rowc + (i64)lk * ldi) * 8u;
            asm volatile("" : "+v"(vo));
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) bf[i][ks] = *reinterpret_cast<const double *>(Pb + (size_t)(4 * ks) * (size_t)ldi * 8u + vo);
            return;
        }
        const i32 wi = min(max(w - 64 * i, 1), NB_IN);                 // columns of step i (>= 1: the address stays in bounds)
        const i32 ci = min(64 * i, w - 1) & ~63;                       // an existing slice for steps beyond w
        const double *Pi = pcol(c, fd, k0 + ci) + rowc;
        const i32 ldi = lda - (k0 + ci);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const i32 col = 4 * ks + lk;
            const double v = Pi[(i64)min(col, wi - 1) * ldi];
            bf[i][ks] = (64 * i + col < w) ? v : 0.0;
        }
    };
    load_step(0);
    load_step(1);
    // The 64 x 64 operand blocks are staged in the order (i=0: Linv_0), (i=1: L_10, Linv_1),
    // (i=2: L_20, L_21, Linv_2), ...; block n+1 is fetched into registers while the matrix cores
    // work on block n (global latency off the workgroup's serial chain).
    double pre[16];
    auto fetch = [&](const int i, const int j) {        // j < i: L[k0+64i.., k0+64j..) ; j == i: Linv_i
        const i32 nbi = min(NB_IN, w - 64 * i);
        if constexpr (fullw) {                                        // thread (cc = lane, k = wave + 4 u): a uniform column, the lane's row
            unsigned vo = (unsigned)lane * 8u;
            asm volatile("" : "+v"(vo));
            if (j < i) {
                const i32 ldj = lda - (k0 + 64 * j);
                const char *Pb = reinterpret_cast<const char *>(pcol(c, fd, k0 + 64 * j) + (k0 + 64 * i));
#pragma unroll
                for (int u = 0; u < 16; ++u) pre[u] = *reinterpret_cast<const double *>(Pb + (size_t)(wave + 4 * u) * (size_t)ldj * 8u + vo);
            } else {
                const char *Wb_ = reinterpret_cast<const char *>(front_dinv(c, fd, k0 + 64 * i));
#pragma unroll
                for (int u = 0; u < 16; ++u) pre[u] = *reinterpret_cast<const double *>(Wb_ + (size_t)(wave + 4 * u) * (size_t)(NB_IN * 8) + vo);
            }
            return;
        }
        if (j < i) {
            const double *Pj = pcol(c, fd, k0 + 64 * j);
            const i32 ldj = lda - (k0 + 64 * j);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int idx = tid + 256 * u, cc = idx & (NB_IN - 1), k = idx >> 6;
                pre[u] = (cc < nbi) ? Pj[(i64)(k0 + 64 * i + cc) + (i64)k * ldj] : 0.0;
            }
        } else {
            const double *W = front_dinv(c, fd, k0 + 64 * i);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int idx = tid + 256 * u, cc = idx & (NB_IN - 1), k = idx >> 6;
                pre[u] = (cc < nbi && k < nbi) ? W[(i64)cc + (i64)k * nbi] : 0.0;
            }
        }
    };
    double *Ws = Wb[0];
    auto stage = [&]() {
        Ws = Wb[wsel]; wsel ^= 1;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = tid + 256 * u, cc = idx & (NB_IN - 1), k = idx >> 6;
            Ws[k * LDW + cc] = pre[u];
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (64 * i < w) {                            // wave-uniform
            if (i + 2 < 4) load_step(i + 2);
            v4f64 acc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < i; ++j) {            // solved steps: acc += X_j * L[k0+64i.., k0+64j..]'
                stage();
                __syncthreads();
                fetch(i, j + 1);
                if (active) {
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                        for (int a = 0; a < 4; ++a)
                            acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[(4 * ks + lk) * LDW + a * 16 + lr], bf[j][ks], acc[a], 0, 0, 0);
                }
            }
            if (i > 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bf[i][4 * a + q] -= acc[a][q];
            }
            stage();
            __syncthreads();
            if (i < 3 && 64 * (i + 1) < w) fetch(i + 1, 0);
            if (active) {
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        if (4 * ks > 16 * a + 15) continue;       // Linv[c][k] = 0 for k > c: whole block is zero
                        acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[(4 * ks + lk) * LDW + a * 16 + lr], bf[i][ks], acc[a], 0, 0, 0);
                    }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bf[i][4 * a + q] = acc[a][q];
                // the step's columns are final: store them now, the later steps of the block column run while the stores drain
                if (rbase + lr < rlim) {
                    if constexpr (fullw) {
                        const i32 ldi = lda - (k0 + 64 * i);
                        char *Pb = reinterpret_cast<char *>(pcol(c, fd, k0 + 64 * i));
                        unsigned vo = (unsigned)((i64)(rbase + lr) + (i64)lk * ldi) * 8u;
                        asm volatile("" : "+v"(vo));
#pragma unroll
                        for (int ks = 0; ks < 16; ++ks)
                            *reinterpret_cast<double *>(Pb + (size_t)(4 * ks) * (size_t)ldi * 8u + vo) = SIGNED ? bf[i][ks] * c.csign[fd.col0 + k0 + 64 * i + 4 * ks + lk] : bf[i][ks];
                    } else {
                    double *Pi = pcol(c, fd, k0 + 64 * i) + rbase + lr;
                    const i32 ldi = lda - (k0 + 64 * i);
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const i32 col = 64 * i + 4 * ks + lk;
                        if (col < w) Pi[(i64)(4 * ks + lk) * ldi] = SIGNED ? bf[i][ks] * c.csign[fd.col0 + k0 + col] : bf[i][ks];
                    }
                    }
                }
            }
        }
    }
}
// ------------------------------------------------------------------------------------------
// Full-width block columns (w == 256), round 6 (last): the ten 64 x 64 operand blocks of a strip -- Linv_0 | L_10 Linv_1 | L_20 L_21 Linv_2 | L_30 L_31 L_32 Linv_3 --
// travel global -> LDS WITHOUT passing the registers (global_load_lds_dwordx4: 64 lanes x 16 bytes = two 64-entry rows per instruction, eight instructions per
// wave and block), into a ring of NBUF buffers, NBUF - 1 blocks ahead of the products.  trsm_task above keeps ONE block in flight in 32 registers and stages it
// with 16 LDS writes per thread: a block's latency had the ~1.7 us of the previous block's products to hide in and needed ~3 (sixty strips pull the same 32 KB
// at the same moment) -- 30 us per strip for 14.5 us of matrix-core time, on the critical path of the dependency-driven launches.  k_trsm: NBUF = 2 (two
// workgroups per CU: the other one covers the latency; the staging writes and the 32 registers go away); k_chain: NBUF = 4 (the CU is the workgroup's own).
// LDS image of a block: Ws[k * 64 + (cc ^ 16 (k & 1))] -- no padding (the hardware writes lane l's 16 bytes at base + 16 l), the XOR puts the rows k, k + 1
// that the lanes lk, lk + 1 of an operand read hit into different halves of the banks, like the stride 80 of the staged form.
// Waiting: the loads are inline assembly, the compiler's counters do not see them; s_waitcnt vmcnt(8 x blocks that may still fly) in front of the barrier of a
// block.  That count is exact because nothing else of this wave is in flight: the strip's own rows are all loaded up front (they are older: loads return in order)
// and ALL stores wait for the end (a store may return before an older load: one of them in the ring would break the count); the kernel must not spill.
// Same products in the same order as trsm_task: identical bits.
// ------------------------------------------------------------------------------------------
#ifndef TLPK_TRSM_DMA
#define TLPK_TRSM_DMA 1
#endif
constexpr bool TRSM_DMA = TLPK_TRSM_DMA != 0;
constexpr int TRSM_DMA_BLK = NB_IN * NB_IN;          // doubles per ring buffer
typedef __attribute__((address_space(3))) double trsm_lds_double;
#pragma clang diagnostic ignored "-Winline-asm"      /* "m0" in the clobber list of the LDS-DMA statements (here and in update_tile) */
template <int PEND> __device__ __forceinline__ void trsm_dma_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PEND) : "memory");
}
// EARLY ENTRY (k_chain, TrsmTask.pad2 > 0): the strip does not wait for the diagonal block of its block column to be COMPLETE.  The diagonal-block role raises
// the block column's counter at three points on its way (prog: 1 = step 0 factored and solved inside the block column, 2, 3 likewise; its final signal makes 4),
// and the strip consumes the operand blocks as they become final -- Linv_0, L_10 at 1; Linv_1, L_20, L_21 at 2; Linv_2, L_30, L_31, L_32 at 3; Linv_3 at 4 --,
// so that behind the diagonal block's last instruction only ONE product (40 of the 544 matrix-core instructions of a strip) and the stores are left: the strips
// were 25 us of the chain's 124 us period, all of them behind the diagonal block.
// Wave 0 polls the counter (one lane, the back-off of chain_wait) and hands the value to the other waves through one LDS word in front of a barrier: every
// wave takes the same decisions, the ring's bookkeeping (`issued`) is workgroup-uniform.  A poll drains the polling wave's ring (its value returns behind the
// older loads): polls only happen while the strip is AHEAD of the diagonal block.  Giving up (time limit, or somebody else did): the word becomes 99, the
// strip runs to its end on whatever the blocks hold -- the host reports TLPK_INTERNAL, nothing hangs.
struct TrsmProg { const unsigned *cnt; int *info; unsigned timeout_ms; };
template <bool SIGNED, int NBUF, bool EARLY>
__device__ __forceinline__ void trsm_task_dma(const TrsmTask t, const FrontDesc &fd, const DevCtx &c, double *b01, double *b23, const TrsmProg pg) {
    static_assert(NBUF == 2 || NBUF == 4, "ring of two (k_trsm) or four (k_chain) buffers");
    constexpr int SI[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3}, SJ[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
    constexpr unsigned long long ITAB = 0x3333222110ull, JTAB = 0x3210210100ull, NTAB = 0x4333322211ull;      // nibble n: step, solved step, progress needed
    const i32 k0 = t.k0;
    const i32 lda = fd.lda;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const i32 rbase = t.row0 + wave * 16;
    const i32 rlim = t.pad1;
    const bool active = rbase < rlim;
    const i32 rowc = min(rbase + lr, rlim - 1);
    double *bp[NBUF];
    unsigned la[NBUF];
#pragma unroll
    for (int q = 0; q < NBUF; ++q) {
        bp[q] = q < 2 ? b01 + q * TRSM_DMA_BLK : b23 + (q - 2) * TRSM_DMA_BLK;
        la[q] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(trsm_lds_double *)bp[q]);
    }
    volatile unsigned *pw = reinterpret_cast<volatile unsigned *>(b01 + 2 * TRSM_DMA_BLK);      // the progress word (behind the two images of the static block)
    constexpr bool early = EARLY;
    double bf[4][16];                               // bf[i][ks] = B[row][k0 + 64 i + 4 ks + lk]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const i32 ldi = lda - (k0 + 64 * i);
        const char *Pb = reinterpret_cast<const char *>(pcol(c, fd, k0 + 64 * i));
        unsigned vo = (unsigned)((i64)rowc + (i64)lk * ldi) * 8u;
        asm volatile("" : "+v"(vo));
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) bf[i][ks] = *reinterpret_cast<const double *>(Pb + (size_t)(4 * ks) * (size_t)ldi * 8u + vo);
    }
    asm volatile("" ::: "memory");                  // the rows' loads stay in front of the ring's
    const unsigned half = (unsigned)lane >> 5;      // lanes 0..31: row k of a pair, 32..63: row k + 1
    const unsigned ccs = (2u * ((unsigned)lane & 31u)) ^ (16u * half);
    auto issue = [&](const int m) {                 // block m of the sequence into buffer m % NBUF (m: wave-uniform)
        const int i = (int)((ITAB >> (4 * m)) & 15u), j = (int)((JTAB >> (4 * m)) & 15u);
        const bool below = j < i;
        const i32 ld = below ? lda - (k0 + 64 * j) : NB_IN;
        const char *gb = below ? reinterpret_cast<const char *>(pcol(c, fd, k0 + 64 * j) + (k0 + 64 * i))        // L[k0 + 64 i + cc][k0 + 64 j + k]
                               : reinterpret_cast<const char *>(front_dinv(c, fd, k0 + 64 * i));                 // Linv_i[cc][k], column-major, ld 64
        unsigned vo = (ccs + half * (unsigned)ld) * 8u;
        asm volatile("" : "+v"(vo));
        const int q = m & (NBUF - 1);
        const unsigned lbase = NBUF == 2 ? (q ? la[1] : la[0]) : (q == 0 ? la[0] : (q == 1 ? la[1] : (q == 2 ? la[NBUF - 2] : la[NBUF - 1])));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const char *sb = gb + (size_t)(2 * (8 * wave + e)) * (size_t)ld * 8u;
            const unsigned m0v = lbase + (unsigned)(8 * wave + e) * 1024u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(vo), "s"(sb) : "memory", "m0");
        }
    };
    unsigned have = early ? 0u : 99u;               // progress of the diagonal block as this wave knows it (workgroup-uniform behind every barrier)
    auto need_of = [&](const int m) { return (unsigned)((NTAB >> (4 * m)) & 15u); };
    auto poll = [&]() {                             // wave 0
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_load(pg.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    };
    auto publish_word = [&](const unsigned v) {     // wave 0: the producers' data first (acquire), then the word
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane == 0) *pw = v;
    };
    int issued = 0;
    unsigned long long t_seen = 0;                  // (wave 0) when `have` last changed, 100 MHz clock
    // operand (k = 4 ks + lk, cc = 16 a + lr) of the image: k & 1 = lk & 1
    int rd[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) rd[a] = lk * NB_IN + ((a * 16 + lr) ^ (16 * (lk & 1)));
    v4f64 acc[4];
#pragma unroll
    for (int n = 0; n < 10; ++n) {
        const int i = SI[n], j = SJ[n];
        if (issued <= n) {                           // (workgroup-uniform) block n is not on its way yet: its operands were not final at the last look
            if (early) {
                if (wave == 0 && have < need_of(n)) {
                    const unsigned long long t0 = wall_clock64();
                    unsigned spins = 0;
                    // Lean polling (the words of a few diagonal blocks are watched by every strip of their block columns -- up to two hundred workgroups --, and a
                    // memory channel flooded with polls starves the workgroups whose data it serves: profiles/r06_chain_poll_storm.txt).  Only the LAST value is
                    // waited for on the critical path, and only by the strips whose rows the next diagonal block needs; it comes no sooner than ~15 us behind the
                    // third.  So: every 3 us for the early values and for the strips further down; the critical strips stay away for 12 us behind the third value
                    // and then look every 0.2 us (64 times), every 0.8 us after that.
                    const bool fast = need_of(n) == 4u && t.row0 < t.k0 + 2 * NB_OUT;      // (workgroup-uniform)
                    if (fast) while (wall_clock64() - t_seen < 1200ull) __builtin_amdgcn_s_sleep(32);
                    for (;;) {
                        have = poll();
                        if (have >= need_of(n)) break;
                        if (!fast) __builtin_amdgcn_s_sleep(127);
                        else if (spins < 64u) __builtin_amdgcn_s_sleep(8);
                        else __builtin_amdgcn_s_sleep(32);
                        if ((++spins & 63u) == 0u) {
                            int flag = 0;
                            if (lane == 0) flag = __hip_atomic_load(pg.info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const bool late = wall_clock64() - t0 > (unsigned long long)pg.timeout_ms * 100000ull;
                            if (__builtin_amdgcn_readfirstlane(flag) != 0 || late) {
                                if (late && lane == 0) __hip_atomic_store(pg.info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                have = 99u;
                                break;
                            }
                        }
                    }
                    t_seen = wall_clock64();
                    publish_word(have);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                have = (unsigned)__builtin_amdgcn_readfirstlane((int)*pw);
            }
            issue(n); ++issued;
        }
        // ahead, into the buffers that are free in front of this block's barrier
        while (issued < 10 && issued <= n + NBUF - 2 && have >= need_of(issued)) { issue(issued); ++issued; }
        if (early && wave == 0 && issued < 10 && have < need_of(issued) && wall_clock64() - t_seen >= 1000ull) {      // one look per block while the diagonal block is behind (not within 10 us of the last change)
            const unsigned v = poll();
            if (v > have) { have = v; t_seen = wall_clock64(); publish_word(have); }
        }
        const int pend = issued - 1 - n;             // blocks behind block n that may still be in flight at its barrier (<= NBUF - 2)
        if (pend <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (pend == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // block n has landed (every wave's share: the barrier), and every wave is done with block n - 1: its buffer is free
        if (early) have = (unsigned)__builtin_amdgcn_readfirstlane((int)*pw);
        while (issued < 10 && issued <= n + NBUF - 1 && have >= need_of(issued)) { issue(issued); ++issued; }
        const double *Ws = bp[n % NBUF];
        if (j == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = (v4f64){0.0, 0.0, 0.0, 0.0};
        }
        if (j < i) {                                 // solved step j: acc += X_j * L[k0 + 64 i.., k0 + 64 j..]'
            if (active) {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[4 * ks * NB_IN + rd[a]], bf[j][ks], acc[a], 0, 0, 0);
            }
            if (j == i - 1) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bf[i][4 * a + q] -= acc[a][q];
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = (v4f64){0.0, 0.0, 0.0, 0.0};
            }
        } else if (active) {                         // X_i = B_i Linv_i'
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (4 * ks > 16 * a + 15) continue;       // Linv[c][k] = 0 for k > c: whole block is zero
                    acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[4 * ks * NB_IN + rd[a]], bf[i][ks], acc[a], 0, 0, 0);
                }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) bf[i][4 * a + q] = acc[a][q];
        }
    }
    if (rbase + lr < rlim) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const i32 ldi = lda - (k0 + 64 * i);
            char *Pb = reinterpret_cast<char *>(pcol(c, fd, k0 + 64 * i));
            unsigned vo = (unsigned)((i64)(rbase + lr) + (i64)lk * ldi) * 8u;
            asm volatile("" : "+v"(vo));
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                *reinterpret_cast<double *>(Pb + (size_t)(4 * ks) * (size_t)ldi * 8u + vo) = SIGNED ? bf[i][ks] * c.csign[fd.col0 + k0 + 64 * i + 4 * ks + lk] : bf[i][ks];
        }
    }
}
static_assert(2 * TRSM_DMA_BLK + 2 <= 2 * NB_IN * LDW, "trsm_task_dma: two ring buffers and the progress word in the staged form's LDS");

template <bool SIGNED>
__global__ __launch_bounds__(256, 2) void k_trsm(const TrsmTask *__restrict__ tasks, DevCtx c) {
    __shared__ __attribute__((aligned(16))) double Wb[2][NB_IN * LDW];           // staged operand: Wb[.][k*LDW + c]; trsm_task_dma: two 64 x 64 images
    const TrsmTask t = tasks[blockIdx.x];
    const FrontDesc fd = c.fronts[t.front];
    if (t.nb == NB_OUT) {                                            // (workgroup-uniform)
        if constexpr (TRSM_DMA) trsm_task_dma<SIGNED, 2, false>(t, fd, c, &Wb[0][0], nullptr, TrsmProg{nullptr, nullptr, 0u});
        else trsm_task<SIGNED, true>(t, fd, c, Wb);
    } else trsm_task<SIGNED, false>(t, fd, c, Wb);
}

// Thin block columns (w <= TRSM_THIN_W: the small fronts of the leaf levels): X = B * L11^{-T} with
// one thread per row -- the row's w entries in registers, the inverted block broadcast from LDS;
// consecutive threads touch consecutive rows of each column (coalesced).
template <bool SIGNED>
__global__ __launch_bounds__(256) void k_trsm_thin(const TrsmTask *__restrict__ tasks, DevCtx c) {
    __shared__ double Wl[TRSM_THIN_W * TRSM_THIN_W];
    const TrsmTask t = tasks[blockIdx.x];
    const FrontDesc fd = c.fronts[t.front];
    const i32 w = t.nb;
    const i32 lda = pld(fd, t.k0);                  // leading dimension of the slice of the packed panel that holds the block column
    const double *W = front_dinv(c, fd, t.k0);          // w x w, column-major, ld = w, upper part zero
    for (int idx = threadIdx.x; idx < w * w; idx += 256) Wl[idx] = W[idx];
    __syncthreads();
    const i32 r = t.row0 + threadIdx.x;
    if (r >= t.pad1) return;                           // rows [row0, pad1) belong to this task
    double *P = pcol(c, fd, t.k0) + r;              // thin block columns (w <= 32) never straddle a 64-column slice: fronts of <= 32 pivot columns, k0 = 0
    double b[TRSM_THIN_W];
#pragma unroll
    for (int k = 0; k < TRSM_THIN_W; ++k) b[k] = P[(i64)min(k, w - 1) * lda];          // clamped, used only for k < w
#pragma unroll
    for (int cc = 0; cc < TRSM_THIN_W; ++cc) {
        if (cc < w) {                                    // workgroup-uniform
            double x = 0.0;
#pragma unroll
            for (int k = 0; k <= cc; ++k) x += Wl[cc + k * w] * b[k];
            P[(i64)cc * lda] = SIGNED ? x * c.csign[fd.col0 + t.k0 + cc] : x;
        }
    }
}

// ------------------------------------------------------------------------------------------
// update: T[I,J] -= P[I,K] * P[J,K]'  on the lower triangle, one TILE x TILE tile per workgroup,
// on the fp64 matrix cores.  v_mfma_f64_16x16x4_f64: A operand lane l = A[l&15][l>>4], B operand
// lane l = B[l>>4][l&15], result reg r of lane l = D[(l>>4)+4r][l&15]
// (/opt/skills/guides/cdna_hip_programming.md section 3).  The MFMA "A" operand is fed from the
// COLUMN tile (rows J of the panel) and "B" from the ROW tile so that D[i][j] = T[I0+j, J0+i]:
// consecutive lanes then hold consecutive rows of one target column -> coalesced column-major
// read-modify-write.
// Targets with column < ns live in the panel, the others in the update matrix U.
// ------------------------------------------------------------------------------------------
#ifndef TLPK_UPD_DMA
#define TLPK_UPD_DMA 1
#endif
constexpr bool UPD_DMA = TLPK_UPD_DMA != 0;       // k_update's slabs global -> LDS directly (update_tile, main path); 0 (build-time, diagnostics): through registers
constexpr int UPD_KT = 16;                 // K depth staged per LDS round
constexpr int UPD_LD = TILE + 16;          // LDS row stride (doubles), == 16 mod 32: conflict-free b64 reads

// FULL = interior tile: all 128 rows of both operand tiles exist, the tile lies strictly below
// the diagonal and inside the column limit, so every 16x16 block is a target and no load needs a
// guard except the K tail.  The hot loop is then straight-line code: 16 unguarded staging loads,
// 4 x (8 LDS reads + 16 MFMAs).  Edge and diagonal tiles take the guarded generic path.
// SIGNED (K2): T -= P[I,K] S_K P[J,K]': the column-tile operand is multiplied by the signs of its K columns while it is staged.
// NW = waves per workgroup.  4: 2 x 2 waves, a 64 x 64 sub-tile each (16 accumulator blocks per wave).  8: 2 x 4 waves, 64 rows x 32
// columns each (8 accumulator blocks, ~half the registers): twice the waves per SIMD to cover LDS / barrier / load waits.
template <bool FULL, bool SIGNED, int NW>
__device__ __forceinline__ void update_tile(const UpdateTask t, const FrontDesc &fd, const DevCtx &c,
                                            double (*As)[UPD_KT * UPD_LD], double (*Bs)[UPD_KT * UPD_LD]) {
    const double *sgk = SIGNED ? c.csign + fd.col0 + t.k0 : nullptr;          // sign of K column k: sgk[k]
    const i32 f = fd.f, ns = fd.ns, rs = f - ns;
    const i32 lda = fd.lda;                         // leading dimension of the panel (>= f, multiple of 16 for large fronts)
    const double *P = c.Lval + fd.loff;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR
    constexpr int CA = (NW == 4) ? 4 : 2;           // 16-column blocks per wave
    constexpr int WCW = 16 * CA;                    // columns of a wave's sub-tile
    constexpr int KS = NW / 2;                      // staging: threads with the same row sr take every KS-th k
    constexpr int UPD_NLD = UPD_KT / KS;            // staging loads per thread per operand per round
    const int wr = (NW == 4) ? (wave >> 1) : (wave >> 2), wc = (NW == 4) ? (wave & 1) : (wave & 3);
    const i32 ibase = t.i0 + wr * 64;               // target rows of this wave
    const i32 jbase = t.j0 + wc * WCW;              // target cols of this wave
    const bool diag_tile = !FULL && !SIGNED && (t.i0 == t.j0);       // unsigned: the column tile IS the row tile (one staging buffer)

    bool valid[CA][4];
    bool any = FULL;
    if constexpr (!FULL) {
#pragma unroll
        for (int a = 0; a < CA; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const i32 cb = jbase + a * 16, rb = ibase + b * 16;   // block columns [cb,cb+16), rows [rb,rb+16)
                valid[a][b] = (rb < f) && (cb < t.jlim) && (rb + 15 >= cb);
                any |= valid[a][b];
            }
    }
    v4f64 acc[CA][4];
#pragma unroll
    for (int a = 0; a < CA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4f64){0.0, 0.0, 0.0, 0.0};

    const int lr = lane & 15, lk = lane >> 4;
    const int sr = tid & 127, sk0 = tid >> 7;       // staging: row sr, k = sk0 + KS*it
    const i32 ra = t.i0 + sr, rb_ = t.j0 + sr;
    const bool raok = FULL || (ra < f), rbok = FULL || ((rb_ < f) && !diag_tile);
    // packed panel (tlpk_host.hpp: pk_off): K column k of the front starts at P + pk_off(lda, k); a 16-column slab lies in one slice.
    // The K column of a staging load (sk0 + KS * it inside the slab) is wave-uniform: its offset is scalar arithmetic.
    const int sk0u = __builtin_amdgcn_readfirstlane(sk0);
    double pa[UPD_NLD], pb[UPD_NLD];

    auto load_slab = [&](i32 kk, bool full_k) {
        if (FULL && full_k) {
#pragma unroll
            for (int it = 0; it < UPD_NLD; ++it) {
                const double *Pk = P + pk_off(lda, t.k0 + kk + sk0u + KS * it);
                pa[it] = Pk[ra];
                pb[it] = Pk[rb_];
                if (SIGNED) pb[it] *= sgk[kk + sk0 + KS * it];
            }
        } else {
#pragma unroll
            for (int it = 0; it < UPD_NLD; ++it) {
                const bool kok = (kk + sk0u + KS * it) < t.kw;
                const double *Pk = P + pk_off(lda, t.k0 + min(kk + sk0u + KS * it, t.kw - 1));
                pa[it] = (kok && raok) ? Pk[ra] : 0.0;
                pb[it] = (kok && rbok) ? Pk[rb_] : 0.0;
                if (SIGNED) pb[it] *= sgk[min(kk + sk0 + KS * it, t.kw - 1)];
            }
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int it = 0; it < UPD_NLD; ++it) {
            const int k = sk0 + KS * it;
            As[buf][k * UPD_LD + sr] = pa[it];
            if (FULL || !diag_tile) Bs[buf][k * UPD_LD + sr] = pb[it];
        }
    };

    // K-segment list (UpdateTask.seg): the slabs in which both operand row ranges hold structural nonzeros; wave-uniform (scalar loads)
    const i32 *segp = t.seg ? c.upd_seg + (t.seg - 1) : nullptr;
    if (segp || t.kw >= 2 * UPD_KT) {
        // Main path (every tile with K >= 2 slabs): the staging work is spread INSIDE the MFMA block
        // instead of in front of it (ablation: the 16 loads + address arithmetic issued before the
        // first MFMA of a round cost ~19 % even with L2-hot data, i.e. pure issue time).  Registers
        // hold slab t+1 at the start of round t: its LDS stores go with the first MFMA group, the
        // loads of slab t+2 with the second and third, so the prefetch distance is two rounds.
        // Edge and diagonal tiles run the same straight-line loop: their staging rows are clamped to
        // the last row of the front (a clamped row only feeds outputs that the guarded epilogue
        // never stores), all 16 MFMA blocks of an active wave are computed, the epilogue masks.
        // One address register pair per staging load, advanced in place: when the addresses were recomputed from a
        // base pointer every round, the compiler recycled the destination registers of loads still in flight as
        // temporaries and had to wait for them (s_waitcnt vmcnt(3) in the middle of a round: the two-round prefetch
        // distance shrank to half a round).
        // Packed panel: the K columns of one staging load (sk0 + KS * it inside the slab) are wave-uniform, so the advance of its
        // address from slab to slab -- UPD_KT columns of lda - 64 b inside slice b, a different amount across a slice boundary -- is
        // scalar arithmetic: pk_off(next column) - pk_off(this column).
        // Addresses = wave-uniform base (scalar registers: the panel + the offset of the K column) + a constant 32-bit byte offset
        // per lane (its clamped row): the advance from slab to slab -- and the jump to the next K segment of a skip list -- is
        // scalar arithmetic only, no address registers per load (the kernel sits at the 128-register limit of 4 waves / SIMD).
        // Both operands read the same K columns; ld_a and ld_b of a round load the SAME slab, the iterator moves after ld_b.
        unsigned voa = (unsigned)min(t.i0 + sr, f - 1) * 8u, vob = (unsigned)min(t.j0 + sr, f - 1) * 8u;
        const char *Pc = reinterpret_cast<const char *>(P);
        const i32 nseg = __builtin_amdgcn_readfirstlane(segp ? segp[0] : 1);
        // A slab of UPD_KT = 16 columns lies inside ONE slice b (the K ranges start on multiples of 16): the next slab is
        // 16 (lda - 64 b) doubles further; when it opens slice b + 1, column j of the slab moves 64 (j + 1) doubles less
        // (pk_off(c + 16) - pk_off(c) for c = 64 b + 48 + j).  Skip lists: when the current K segment is used up the iterator
        // jumps to the first column of the next one (offsets from pk_off again).
        i32 k_slab = __builtin_amdgcn_readfirstlane(segp ? segp[1] : t.k0);               // first K column of the slab loaded next
        i32 k_rem = __builtin_amdgcn_readfirstlane(segp ? segp[2] : (t.kw / UPD_KT));     // slabs left in the current segment
        i32 k_seg = 0;
        const char *sb[UPD_NLD];                          // address of (row 0 of) K column k_slab + sk0u + KS * it (scalar)
#pragma unroll
        for (int it = 0; it < UPD_NLD; ++it) sb[it] = Pc + pk_off(lda, k_slab + sk0u + KS * it) * 8;
        i64 step = (i64)UPD_KT * 8 * (lda - ((k_slab >> 6) << 6));       // 16 columns inside the current slice, in bytes
        // (the empty asm keeps the zero-extension of the lane offset in the block of the load: hoisted out of the loop, instruction selection
        // no longer sees "scalar base + 32-bit lane offset" and goes back to a 64-bit vector add per load, whose temporaries are the
        // destination registers of loads still in flight)
        auto ld_a = [&]() {
            asm volatile("" : "+v"(voa));               // in place: a copy would land in a destination register of the previous loads
#pragma unroll
            for (int it = 0; it < UPD_NLD; ++it) pa[it] = *reinterpret_cast<const double *>(sb[it] + voa);
        };
        const double *sgf = SIGNED ? c.csign + fd.col0 : nullptr;          // sign of the front's K column k: sgf[k]
        auto ld_b = [&]() {
            asm volatile("" : "+v"(vob));
#pragma unroll
            for (int it = 0; it < UPD_NLD; ++it) {
                pb[it] = *reinterpret_cast<const double *>(sb[it] + vob);
                if (SIGNED) pb[it] *= sgf[k_slab + sk0 + KS * it];
            }
            k_slab += UPD_KT;
            // common path without a taken branch (a slab round is ~4 000 cycles, a taken branch refills the instruction buffer): the
            // slice-boundary fix-up -- every fourth slab -- is a scalar select, the end of a K segment the only branch
            const bool cross = (k_slab & 63) == 0;        // (wave-uniform) the next slab opens a new slice
#pragma unroll
            for (int it = 0; it < UPD_NLD; ++it) sb[it] += step - (cross ? (i64)(64 * 8) * (sk0u + KS * it + 1) : 0);
            step -= cross ? 64 * 8 * UPD_KT : 0;
            if (__builtin_expect(--k_rem == 0, 0)) {      // (wave-uniform) end of the K segment: jump to the next one, if any
                if (++k_seg < nseg) {
                    k_slab = __builtin_amdgcn_readfirstlane(segp[1 + 2 * k_seg]);
                    k_rem = __builtin_amdgcn_readfirstlane(segp[2 + 2 * k_seg]);
#pragma unroll
                    for (int it = 0; it < UPD_NLD; ++it) sb[it] = Pc + pk_off(lda, k_slab + sk0u + KS * it) * 8;
                    step = (i64)UPD_KT * 8 * (lda - ((k_slab >> 6) << 6));
                }
            }
        };
        auto st_ab = [&](int buf) {
#pragma unroll
            for (int it = 0; it < UPD_NLD; ++it) {
                As[buf][(sk0 + KS * it) * UPD_LD + sr] = pa[it];
                Bs[buf][(sk0 + KS * it) * UPD_LD + sr] = pb[it];
            }
        };
        auto mfma_round = [&](int buf, auto &&hook) {
            const double *At = As[buf] + wr * 64 + lr + lk * UPD_LD;
            const double *Bt = Bs[buf] + wc * WCW + lr + lk * UPD_LD;
#pragma unroll
            for (int k4 = 0; k4 < UPD_KT; k4 += 4) {
                double av[CA], bv[4];
                if (any) {
#pragma unroll
                    for (int a = 0; a < CA; ++a) av[a] = Bt[k4 * UPD_LD + a * 16];
#pragma unroll
                    for (int b = 0; b < 4; ++b) bv[b] = At[k4 * UPD_LD + b * 16];
                }
                hook(k4);
                if (any) {
#pragma unroll
                    for (int a = 0; a < CA; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
                }
            }
        };
        int cur = 0;
        const i32 nrounds = segp ? t.nsl : t.kw / UPD_KT;
        // (16-byte requests, and the LDS-DMA path wants them ALIGNED -- a misaligned one does not fault, it delivers the wrong rows --: the panel, its leading dimension
        //  and the tile's first row and column must be even.  Panels of fronts with >= 64 rows start on 128-byte lines with an even lda; tiles inside the pivot columns
        //  start on multiples of 128; the tiles of the update matrix start at ns: those of a front with an odd number of pivot columns keep the registers)
        const bool dma_ok = UPD_DMA && !SIGNED && ((fd.loff | (i64)lda | (i64)t.i0 | (i64)t.j0) & 1) == 0;      // (workgroup-uniform)
        if (dma_ok) {
            // Round 6 (last): the slabs travel global -> LDS WITHOUT passing the registers (global_load_lds_dwordx4: a K column of a tile -- 128 rows = 1 024 bytes -- is ONE
            // instruction of one wave; 16 + 16 columns per slab = four instructions per wave instead of eight loads, eight ds_write_b64 and the wait between them per
            // THREAD).  An ablation without the staging (wrong numbers, valid timing) ran the C4 update 6 % faster: the LDS writes and their waits inside the
            // matrix-core block were the largest single cost left in this kernel.  Slab rd + 1 is requested at the start of round rd, into the buffer every wave has
            // finished reading at the barrier in front of it, and has the whole round (~3 us with four waves per SIMD) to land; s_waitcnt vmcnt(0) in front of the round's
            // closing barrier (nothing else of the wave is in flight inside the loop).  Lane l carries rows 2 l, 2 l + 1 of its column; rows beyond the front are
            // clamped inside the column's padding (lda is even and >= f): they feed outputs the epilogue masks.  Same LDS image, same products: identical bits.
            // (SIGNED tiles multiply the column operand by the signs while staging: they stay on the register path.  The 4-wave role of k_chain -- one wave per
            // SIMD, a round of 1.7 us to hide the slab in -- gains as well: pds-class 12.8 -> 12.5 ms, rank-local N = 8 11.1 -> 10.8 ms.)
            typedef __attribute__((address_space(3))) double upd_lds_double;
            const unsigned la_a0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(upd_lds_double *)&As[0][0]);
            const unsigned la_b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(upd_lds_double *)&Bs[0][0]);
            constexpr unsigned BUFB = UPD_KT * UPD_LD * 8;                 // bytes per buffer
            constexpr int CPW = UPD_KT / NW;                               // K columns of a slab per wave
            unsigned dva = (unsigned)min(t.i0 + 2 * lane, lda - 2) * 8u, dvb = (unsigned)min(t.j0 + 2 * lane, lda - 2) * 8u;
            const char *db[CPW];                                           // address of (row 0 of) K column k_slab + wave + NW * e (scalar)
#pragma unroll
            for (int e = 0; e < CPW; ++e) db[e] = Pc + pk_off(lda, k_slab + wave + NW * e) * 8;
            auto dma_slab = [&](const int buf) {
                asm volatile("" : "+v"(dva), "+v"(dvb));
#pragma unroll
                for (int e = 0; e < CPW; ++e) {
                    const unsigned ko = (unsigned)(wave + NW * e) * (unsigned)(UPD_LD * 8) + (unsigned)buf * BUFB;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(la_a0 + ko), "v"(dva), "s"(db[e]) : "memory", "m0");
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(la_b0 + ko), "v"(dvb), "s"(db[e]) : "memory", "m0");
                }
                k_slab += UPD_KT;
                const bool cross = (k_slab & 63) == 0;        // (wave-uniform) the next slab opens a new slice
#pragma unroll
                for (int e = 0; e < CPW; ++e) db[e] += step - (cross ? (i64)(64 * 8) * (wave + NW * e + 1) : 0);
                step -= cross ? 64 * 8 * UPD_KT : 0;
                if (__builtin_expect(--k_rem == 0, 0)) {      // (wave-uniform) end of the K segment: jump to the next one, if any
                    if (++k_seg < nseg) {
                        k_slab = __builtin_amdgcn_readfirstlane(segp[1 + 2 * k_seg]);
                        k_rem = __builtin_amdgcn_readfirstlane(segp[2 + 2 * k_seg]);
#pragma unroll
                        for (int e = 0; e < CPW; ++e) db[e] = Pc + pk_off(lda, k_slab + wave + NW * e) * 8;
                        step = (i64)UPD_KT * 8 * (lda - ((k_slab >> 6) << 6));
                    }
                }
            };
            dma_slab(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (i32 rd = 0; rd < nrounds; ++rd) {
                const bool have_next = rd + 1 < nrounds;
                mfma_round(cur, [&](int k4) {
                    if (k4 == 0 && have_next) dma_slab(cur ^ 1);
                });
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                cur ^= 1;
            }
        } else {
        ld_a(); ld_b();
        st_ab(0);
        ld_a(); ld_b();                                   // slab 1 in flight
        __syncthreads();
#ifdef UPD_TRACE
        if (threadIdx.x == 0) ((unsigned long long *)c.spart)[(size_t)blockIdx.x * 8 + 1] = wall_clock64();
#endif
        for (i32 rd = 0; rd < nrounds; ++rd) {
            const bool have_next = rd + 1 < nrounds, have_next2 = rd + 2 < nrounds;
            mfma_round(cur, [&](int k4) {
                if (k4 == 0 && have_next) st_ab(cur ^ 1);
                if (k4 == 4 && have_next2) ld_a();
                if (k4 == 8 && have_next2) ld_b();
            });
            __syncthreads();
            cur ^= 1;
        }
        }
        if (t.kw % UPD_KT) {                              // K tail: one zero-filled slab
            const i32 kk = (t.kw / UPD_KT) * UPD_KT;
#pragma unroll
            for (int it = 0; it < UPD_NLD; ++it) {
                const bool kok = (kk + sk0u + KS * it) < t.kw;
                const double *Pk = P + pk_off(lda, t.k0 + min(kk + sk0u + KS * it, t.kw - 1));
                pa[it] = kok ? *reinterpret_cast<const double *>(reinterpret_cast<const char *>(Pk) + voa) : 0.0;
                pb[it] = kok ? *reinterpret_cast<const double *>(reinterpret_cast<const char *>(Pk) + vob) : 0.0;
                if (SIGNED) pb[it] *= sgk[min(kk + sk0 + KS * it, t.kw - 1)];
            }
            st_ab(cur);
            __syncthreads();
            mfma_round(cur, [](int) {});
        }
        goto epilogue;
    }
    {
    load_slab(0, UPD_KT <= t.kw);
    store_slab(0);
    __syncthreads();
    int cur = 0;
    for (i32 kk = 0; kk < t.kw; kk += UPD_KT) {
        const bool more = (kk + UPD_KT) < t.kw;
#if defined(UPD_VARIANT) && (UPD_VARIANT == 1 || UPD_VARIANT == 6)      /* ablation: no global loads in the loop */
        if (more && kk < 0) load_slab(kk + UPD_KT, true);
#else
        if (more) load_slab(kk + UPD_KT, (kk + 2 * UPD_KT) <= t.kw);   // in flight during the MFMA block
#endif
        if (any) {
            const double *At = As[cur] + wr * 64 + lr + lk * UPD_LD;
            const double *Bt = (diag_tile ? As[cur] : Bs[cur]) + wc * WCW + lr + lk * UPD_LD;
#pragma unroll
            for (int k4 = 0; k4 < UPD_KT; k4 += 4) {
                double av[CA], bv[4];
#pragma unroll
                for (int a = 0; a < CA; ++a) av[a] = Bt[k4 * UPD_LD + a * 16];   // column-tile rows
#pragma unroll
                for (int b = 0; b < 4; ++b) bv[b] = At[k4 * UPD_LD + b * 16];   // row-tile rows
#pragma unroll
                for (int a = 0; a < CA; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if constexpr (FULL) {
                            acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
                        } else {
                            if (valid[a][b]) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
                        }
                    }
            }
        }
        if (more) store_slab(cur ^ 1);
#if !(defined(UPD_VARIANT) && (UPD_VARIANT == 4 || UPD_VARIANT == 6))   /* ablation 4/6: no barrier in the loop */
        __syncthreads();
#endif
        cur ^= 1;
    }
    }
epilogue:
#ifdef UPD_TRACE
    if (threadIdx.x == 0) ((unsigned long long *)c.spart)[(size_t)blockIdx.x * 8 + 2] = wall_clock64();
#endif
    if (!any) return;
    if (t.pad1) {
        // split-K part: the raw tile goes to its scratch slot, k_update_reduce applies the parts in order
        double *Sp = c.spart + (i64)(t.pad1 - 1) * (TILE * TILE);
#pragma unroll
        for (int a = 0; a < CA; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if constexpr (!FULL) { if (!valid[a][b]) continue; }
                const i32 lrow = wr * 64 + b * 16 + lr;
#pragma unroll
                for (int q = 0; q < 4; ++q) Sp[lrow + (wc * WCW + a * 16 + lk + 4 * q) * TILE] = acc[a][b][q];
            }
        return;
    }
#if defined(UPD_VARIANT) && (UPD_VARIANT == 2 || UPD_VARIANT == 6)   /* ablation: no epilogue read-modify-write */
#pragma unroll
    for (int a = 0; a < CA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(acc[a][b]));
    return;
#endif
    // epilogue: D[i][j] (reg q: i = lk + 4q, j = lr) = sum_k P[jbase+16a+i, k] * P[ibase+16b+j, k]
    double *Pw = c.Lval + fd.loff;
    double *Uw = front_u(c, fd);
#pragma unroll
    for (int a = 0; a < CA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if constexpr (!FULL) { if (!valid[a][b]) continue; }
            const i32 row = ibase + b * 16 + lr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const i32 col = jbase + a * 16 + lk + 4 * q;
                if (FULL || (row < f && col < t.jlim && row >= col)) {
                    // T -= acc as a fire-and-forget fp64 add executed by the L2 (one adder per entry and launch:
                    // the same IEEE sum as load / subtract / store, bit for bit, without the wave waiting for
                    // the old value -- the read-modify-write form cost 46 us per tile whatever K, its loads
                    // serialised behind the stores to the same array)
                    if (col < ns) unsafeAtomicAdd(Pw + (i64)row + pk_off(lda, col), -acc[a][b][q]);
                    else {
                        double *dst = Uw + (i64)(row - ns) + (i64)(col - ns) * rs;
                        if (t.beta0) *dst = -acc[a][b][q];
                        else unsafeAtomicAdd(dst, -acc[a][b][q]);
                    }
                }
            }
        }
}


// ------------------------------------------------------------------------------------------
// A 64 x 64 update tile per 4-wave workgroup (round 6): the diagonal-block tiles of the dependency-driven launches (UpdateTask.pad2 = 1).  Behind a solved block
// column only K = 256 columns are left to apply to the next 256 x 256 diagonal block, and that update sits ON the chain potrf -> strips -> update -> potrf:
// three 128 x 128 tiles took ~50 us there, the ten 64 x 64 tiles of the block's lower triangle run side by side on ten CUs.  Wave (wr, wc) owns the 32 x 32
// sub-tile: 2 x 2 accumulator blocks, the operand conventions of update_tile (D[i][j] = T[I0 + j, J0 + i]: coalesced column-major targets).  Every entry sums
// its K columns in the order of the 128 x 128 tile -- slab by slab, four columns per v_mfma_f64_16x16x4_f64 -- so the result is the same bit for bit.
// Plain K range or K-segment list; no split-K parts (the schedule gives this shape to K <= 256 only).
// ------------------------------------------------------------------------------------------
constexpr int U64 = 64;                      // tile edge
constexpr int U64_LD = U64 + 16;             // LDS row stride (doubles), == 16 mod 32: conflict-free b64 reads
template <bool SIGNED>
__device__ __forceinline__ void update_tile64(const UpdateTask t, const FrontDesc &fd, const DevCtx &c, double *lds) {
    double (*As)[UPD_KT * U64_LD] = reinterpret_cast<double (*)[UPD_KT * U64_LD]>(lds);          // As[buf][k][row]: rows i0 .. of the panel (row tile)
    double (*Bs)[UPD_KT * U64_LD] = As + 2;                                                       // Bs[buf][k][row]: rows j0 .. (column tile)
    const i32 f = fd.f, ns = fd.ns, rs = f - ns, lda = fd.lda;
    const double *P = c.Lval + fd.loff;
    const double *sgf = SIGNED ? c.csign + fd.col0 : nullptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lk = lane >> 4;
    const i32 ibase = t.i0 + wr * 32, jbase = t.j0 + wc * 32;
    const int sr = tid & 63;
    const int sk0 = __builtin_amdgcn_readfirstlane(tid >> 6);                                     // staging: row sr, K columns sk0 + 4 it of the slab (wave-uniform)
    const i64 ra = min(t.i0 + sr, f - 1), rb = min(t.j0 + sr, f - 1);                            // clamped, the epilogue masks
    v4f64 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const i32 *segp = t.seg ? c.upd_seg + (t.seg - 1) : nullptr;
    const i32 nseg = segp ? segp[0] : 1;
    const i32 nfull = t.kw / UPD_KT, ktail = t.kw % UPD_KT;
    // slab iterator over the (first column, slabs) pairs of the K-segment list, or the one plain range; a partial last slab follows
    i32 seg_i = 0, k_slab = segp ? segp[1] : t.k0, k_rem = segp ? segp[2] : nfull;
    const i32 nrounds = (segp ? t.nsl : nfull) + (ktail ? 1 : 0);
    double pa[4], pb[4];
    auto load_slab = [&](const bool tail) {
        const i32 kbase = tail ? t.k0 + nfull * UPD_KT : k_slab;
        const i32 klim = tail ? ktail : UPD_KT;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const i32 kk = sk0 + 4 * it;
            const double *Pk = P + pk_off(lda, kbase + min(kk, klim - 1));
            const double va = Pk[ra], vb = Pk[rb];
            pa[it] = (kk < klim) ? va : 0.0;
            pb[it] = (kk < klim) ? (SIGNED ? vb * sgf[kbase + min(kk, klim - 1)] : vb) : 0.0;
        }
        if (!tail) {
            k_slab += UPD_KT;
            if (--k_rem == 0 && ++seg_i < nseg) { k_slab = segp[1 + 2 * seg_i]; k_rem = segp[2 + 2 * seg_i]; }
        }
    };
    auto store_slab = [&](const int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) { As[buf][(sk0 + 4 * it) * U64_LD + sr] = pa[it]; Bs[buf][(sk0 + 4 * it) * U64_LD + sr] = pb[it]; }
    };
    const i32 nfull_rounds = nrounds - (ktail ? 1 : 0);
    if (nrounds > 0) {
        load_slab(nfull_rounds == 0);
        store_slab(0);
        __syncthreads();
        int cur = 0;
        for (i32 rd = 0; rd < nrounds; ++rd) {
            const bool more = rd + 1 < nrounds;
            if (more) load_slab(rd + 1 >= nfull_rounds);                                         // in flight during the products
            const double *At = As[cur] + wr * 32 + lr + lk * U64_LD;
            const double *Bt = Bs[cur] + wc * 32 + lr + lk * U64_LD;
#pragma unroll
            for (int k4 = 0; k4 < UPD_KT; k4 += 4) {
                double av[2], bv[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) av[a] = Bt[k4 * U64_LD + a * 16];
#pragma unroll
                for (int b = 0; b < 2; ++b) bv[b] = At[k4 * U64_LD + b * 16];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
            if (more) store_slab(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    // epilogue: as update_tile (fire-and-forget L2 adds; one adder per entry at a time: the schedule orders the adders of a target)
    double *Pw = c.Lval + fd.loff;
    double *Uw = front_u(c, fd);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const i32 row = ibase + b * 16 + lr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const i32 col = jbase + a * 16 + lk + 4 * q;
                if (row < f && col < t.jlim && row >= col && row < t.i0 + U64 && col < t.j0 + U64) {
                    if (col < ns) unsafeAtomicAdd(Pw + (i64)row + pk_off(lda, col), -acc[a][b][q]);
                    else {
                        double *dst = Uw + (i64)(row - ns) + (i64)(col - ns) * rs;
                        if (t.beta0) *dst = -acc[a][b][q];
                        else unsafeAtomicAdd(dst, -acc[a][b][q]);
                    }
                }
            }
        }
}

// ------------------------------------------------------------------------------------------
// A 32 x 32 update tile per workgroup, one 16 x 16 block per WAVE (round 6, UpdateTask.pad2 = 2): the diagonal-block tiles of the dependency-driven launches.
// The K = 256 update of the next diagonal block is a link of the chain (strips -> update -> diagonal block): as 64 x 64 tiles it cost ~28 us, of which 7 were
// matrix-core time -- sixteen slabs, each a global load, an LDS round and a barrier.  Here a wave reads its operands straight from the panel (lane (lr, lk):
// rows ib + lr and jb + lr, column 4 u + lk of a slab: 16 consecutive rows = one 128-byte line per column), eight slabs -- 64 loads -- in flight at a time,
// no LDS, no barrier; the blocks above the diagonal are not formed.  36 tiles per diagonal block instead of ten.
// Every entry sums its K columns in the order of the 128 x 128 tile -- slab by slab, four columns per v_mfma_f64_16x16x4_f64 -- the same bits.
// ------------------------------------------------------------------------------------------
template <bool SIGNED>
__device__ __forceinline__ void update_tile32(const UpdateTask t, const FrontDesc &fd, const DevCtx &c) {
    constexpr int CH = 8;                                                                         // slabs in flight
    const i32 f = fd.f, ns = fd.ns, rs = f - ns, lda = fd.lda;
    const double *P = c.Lval + fd.loff;
    const double *sgf = SIGNED ? c.csign + fd.col0 : nullptr;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lk = lane >> 4;
    const i32 ib = t.i0 + wr * 16, jb = t.j0 + wc * 16;
    if (ib + 15 < jb || ib >= f || jb >= t.jlim) return;                                          // (wave-uniform) above the diagonal / outside: nothing to form
    const i64 ra = min(ib + lr, f - 1), rb = min(jb + lr, f - 1);                                 // clamped, the epilogue masks
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    const i32 *segp = t.seg ? c.upd_seg + (t.seg - 1) : nullptr;
    const i32 nseg = segp ? segp[0] : 1;
    const i32 nfull = t.kw / UPD_KT, ktail = t.kw % UPD_KT;
    i32 seg_i = 0, k_slab = segp ? segp[1] : t.k0, k_rem = segp ? segp[2] : nfull;
    const i32 nfull_rounds = segp ? t.nsl : nfull;
    const i32 nrounds = nfull_rounds + (ktail ? 1 : 0);
    for (i32 rd0 = 0; rd0 < nrounds; rd0 += CH) {
        double pa[CH][4], pb[CH][4];
#pragma unroll
        for (int sl = 0; sl < CH; ++sl)
            if (rd0 + sl < nrounds) {
                const bool tail = rd0 + sl >= nfull_rounds;
                const i32 kbase = tail ? t.k0 + nfull * UPD_KT : k_slab;
                const i32 klim = tail ? ktail : UPD_KT;
                if (!tail) {
                    // a full slab lies inside ONE 64-column slice of the packed panel: column kbase + kk starts kk (lda - 64 b) doubles behind column kbase -- a
                    // wave-uniform base per fragment + one 32-bit lane offset per operand (row + lk columns) instead of a 64-bit address per lane and load
                    const i32 ldb = lda - ((kbase >> 6) << 6);
                    const char *Pb = reinterpret_cast<const char *>(P + pk_off(lda, kbase));
                    unsigned voa = (unsigned)(ra + (i64)lk * ldb) * 8u, vob = (unsigned)(rb + (i64)lk * ldb) * 8u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const char *Pu = Pb + (size_t)(4 * u) * (size_t)ldb * 8u;
                        pa[sl][u] = *reinterpret_cast<const double *>(Pu + voa);
                        const double vb = *reinterpret_cast<const double *>(Pu + vob);
                        pb[sl][u] = SIGNED ? vb * sgf[kbase + 4 * u + lk] : vb;
                    }
                } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const i32 kk = 4 * u + lk, col = kbase + min(kk, klim - 1);
                    const double *Pk = P + pk_off(lda, col);
                    const double va = Pk[ra], vb = Pk[rb];
                    pa[sl][u] = (kk < klim) ? va : 0.0;
                    pb[sl][u] = (kk < klim) ? (SIGNED ? vb * sgf[col] : vb) : 0.0;
                }
                }
                if (!tail) {
                    k_slab += UPD_KT;
                    if (--k_rem == 0 && ++seg_i < nseg) { k_slab = segp[1 + 2 * seg_i]; k_rem = segp[2 + 2 * seg_i]; }
                }
            }
#pragma unroll
        for (int sl = 0; sl < CH; ++sl)
            if (rd0 + sl < nrounds) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[sl][u], pa[sl][u], acc, 0, 0, 0);
            }
    }
    // epilogue: as update_tile64 (fire-and-forget L2 adds; one adder per entry at a time: the schedule orders the adders of a target)
    double *Pw = c.Lval + fd.loff;
    double *Uw = front_u(c, fd);
    const i32 row = ib + lr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const i32 col = jb + lk + 4 * q;
        if (row < f && col < t.jlim && row >= col) {
            if (col < ns) unsafeAtomicAdd(Pw + (i64)row + pk_off(lda, col), -acc[q]);
            else {
                double *dst = Uw + (i64)(row - ns) + (i64)(col - ns) * rs;
                if (t.beta0) *dst = -acc[q];
                else unsafeAtomicAdd(dst, -acc[q]);
            }
        }
    }
}

template <bool SIGNED, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void k_update(const UpdateTask *__restrict__ tasks, DevCtx c) {
    // double-buffered K-slabs: slab t+1 travels global -> registers while slab t is multiplied
    __shared__ __attribute__((aligned(16))) double As[2][UPD_KT * UPD_LD];  // As[.][k][r] = P[i0 + r, k0 + kk + k]  (row tile)
    __shared__ __attribute__((aligned(16))) double Bs[2][UPD_KT * UPD_LD];  // Bs[.][k][r] = P[j0 + r, k0 + kk + k]  (column tile)
    // Workgroups are dealt to the 8 XCDs round-robin by blockIdx.  upd_remap = 2: XCD x takes runs of 64
    // consecutive tasks (neighbours in the list share panel slabs: one L2 serves them); 1: one contiguous
    // eighth of the list per XCD.
    unsigned b = blockIdx.x;
    if (c.upd_remap == 2) {
        const unsigned whole = gridDim.x & ~511u;
        if (b < whole) b = (b & ~511u) + ((b & 7u) << 6) + ((b >> 3) & 63u);
    } else if (c.upd_remap == 1) {
        const unsigned per = gridDim.x >> 3, whole = per << 3;
        if (b < whole) b = (b & 7u) * per + (b >> 3);
    }
    // memberwise: a 48-byte struct copy went through scratch memory (and every wave-uniform field of the task into vector registers)
    const UpdateTask *tp = tasks + b;
    const UpdateTask t{tp->front, tp->k0, tp->kw, tp->i0, tp->j0, tp->jlim, tp->beta0, tp->pad1, tp->seg, tp->nsl, 0, 0};
    const FrontDesc fd = c.fronts[t.front];
    const bool full = (t.i0 + TILE <= fd.f) && (t.j0 + TILE <= t.jlim) && (t.i0 >= t.j0 + TILE);
#ifdef UPD_TRACE   /* tools/update_bench.hip: per-workgroup time stamps (100 MHz) and placement */
    unsigned long long *tr = (unsigned long long *)c.spart + (size_t)blockIdx.x * 8;
    if (threadIdx.x == 0) {
        tr[0] = wall_clock64();
        tr[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
        tr[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20);       // XCC_ID
    }
#endif
    if (full) update_tile<true, SIGNED, NW>(t, fd, c, As, Bs);
    else update_tile<false, SIGNED, NW>(t, fd, c, As, Bs);
#ifdef UPD_TRACE
    if (threadIdx.x == 0) tr[3] = wall_clock64();
#endif
}

// The last partial round of an update launch as 64 x 64 tiles (LK_UPDATE_T64, symbolic.cpp: emit_update_launch): one update_tile64 per 4-wave workgroup, 40 KB of
// LDS and at most 128 registers -- four workgroups per CU, the 16 waves per CU of the 128 x 128 kernel.
template <bool SIGNED>
__global__ __launch_bounds__(256, 4) void k_update64(const UpdateTask *__restrict__ tasks, DevCtx c) {
    __shared__ double lds[4 * UPD_KT * U64_LD];
    const UpdateTask *tp = tasks + blockIdx.x;
    const UpdateTask t{tp->front, tp->k0, tp->kw, tp->i0, tp->j0, tp->jlim, tp->beta0, tp->pad1, tp->seg, tp->nsl, tp->pad2, 0};
    const FrontDesc fd = c.fronts[t.front];
    update_tile64<SIGNED>(t, fd, c, lds);
}

// Split-K: when an update launch has too few tiles to fill the chip, the K range of every tile is
// cut into parts computed by different workgroups (k_update with pad1 != 0 writes the raw
// 128 x 128 partial products to scratch); this kernel adds the parts of one tile in fixed order
// and applies the sum to the tile's targets with the masks of the ordinary epilogue.
// RED_SPLIT workgroups per tile (TILE / RED_SPLIT columns each): a launch that needed split-K has few tiles by
// definition, one workgroup per tile pulled parts x 128 KB through ONE CU (105 us per launch on a 7 900-row front).
__device__ __forceinline__ void reduce_part(const UpdateTask t, const int part, const FrontDesc &fd, const DevCtx &c) {
    const i32 f = fd.f, ns = fd.ns, rs = f - ns;
    const i32 lda = fd.lda;                         // leading dimension of the panel (>= f, multiple of 16 for large fronts)
    const double *Sp = c.spart + (i64)t.k0 * (TILE * TILE);
    double *Pw = c.Lval + fd.loff;
    double *Uw = front_u(c, fd);
    constexpr int PER = TILE * TILE / RED_SPLIT;
    for (int e = part * PER + threadIdx.x; e < (part + 1) * PER; e += 256) {
        const i32 row = t.i0 + (e & (TILE - 1)), col = t.j0 + (e >> 7);
        if (row >= f || col >= t.jlim || row < col) continue;
        double sum = Sp[e];
        for (i32 sp = 1; sp < t.kw; ++sp) sum += Sp[(i64)sp * (TILE * TILE) + e];
        if (col < ns) Pw[(i64)row + pk_off(lda, col)] -= sum;
        else {
            double *dst = Uw + (i64)(row - ns) + (i64)(col - ns) * rs;
            *dst = t.beta0 ? -sum : (*dst - sum);
        }
    }
}
__global__ __launch_bounds__(256) void k_update_reduce(const UpdateTask *__restrict__ tasks, DevCtx c) {
    const UpdateTask t = tasks[blockIdx.x / RED_SPLIT];
    const FrontDesc fd = c.fronts[t.front];
    reduce_part(t, (int)(blockIdx.x % RED_SPLIT), fd, c);
}

// ------------------------------------------------------------------------------------------
// k_chain (round 6): the blocked factorisation of a level's multi-block-column fronts as ONE persistent, dependency-driven launch.
// symbolic.cpp (build_schedule: build_chain) turns the tasks of the launches above -- update tiles, diagonal blocks, 64-row strips of the
// triangular solves, eighths of split-K reductions -- into ITEMS.  A workgroup draws the next item from the launch's ticket counter, waits until
// the completion counters the item names have reached their values, runs the task's ordinary device function (update_tile, potrf_wide_task,
// trsm_task, reduce_part: the arithmetic of the launch form, bit for bit) and raises the item's counter.
//   * visibility (cdna_hip_programming.md, Guideline 16, counter form): the per-die L2s and the per-CU L1s are not coherent for ordinary stores
//     inside a launch.  Producer: every wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane: agent-scope release fence (L2
//     write-back) -> wait -> relaxed agent-scope add on the counter.  Consumer: ONE wave polls its counters with relaxed agent-scope loads, ONE
//     agent-scope acquire fence after the match (drops the stale lines of this CU's L1 / this die's L2), workgroup barrier, then plain loads.
//     The update tiles' fire-and-forget L2 adds are covered by the same release; adders of one target tile are ordered through its counter.
//   * no deadlock by construction: every wait names counters raised by items with SMALLER tickets, whose holders already run; nothing depends
//     on residency, dispatch order or placement (a profiler that serialises dispatches changes nothing: it is one launch).
//   * every spin is bounded (~2 s): the wave that gives up sets info[1], everybody stops waiting and skips the work, the host reports
//     TLPK_INTERNAL.  Counters and tickets are zeroed by one hipMemsetAsync at the start of every update! (no state survives a launch).
// 256 threads, two workgroups per CU (the strips' two staging blocks fill half the LDS): the diagonal block of a front shares its CU with ONE
// other workgroup instead of the four waves per SIMD of the k_update launches it used to run beside.
// ------------------------------------------------------------------------------------------
struct ChainArgs {
    const ChainItem *items; i32 nitems;
    unsigned *cnt; i32 ticket;                       // counters of all chain launches; index of this launch's ticket
    const UpdateTask *upd, *red; const PotrfTask *potrf; const TrsmTask *trsm;
    unsigned long long *trace;                       // TLPK_CHAIN_TRACE=1 (diagnostics): per item 4 words -- drawn, released by its counters, work done, published (100 MHz clock)
    unsigned spin_max;                               // milliseconds before a waiting wave gives up (TLPK_CHAIN_TIMEOUT_MS, default 2000)
    unsigned dyn_lds;                                // bytes of dynamic LDS of the launch (one workgroup per CU: 70 000; the strips' ring of four operand images uses 65 536 of them)
};
constexpr int CHAIN_LDS = 2 * NB_IN * LDW;           // doubles (81 920 bytes): the strips' two staging blocks >= the four K slabs of an update tile, >= the diagonal block's scratch
static_assert(CHAIN_LDS >= 4 * UPD_KT * UPD_LD && CHAIN_LDS >= POTRF_WIDE_DPP_LDS, "k_chain: one LDS block serves every role");

// wave 0 of the workgroup: wait for the item's counters; returns false if somebody (maybe this wave) gave up.
// ONLY the lanes that have a counter to watch poll (at most 3 + the strips of an operand range), and the polls back off: 0.2 us apart for the first 64 -- the
// links of the chain hand over within microseconds --, then 0.8 us, then 3 us.  (Round 6, found by the eight-shards-on-one-GPU test: every one of the 64
// lanes used to load the entry min(lane, ntot - 1), i.e. ~60 duplicate agent-scope loads of ONE word per poll and waiting workgroup; with 200 workgroups
// waiting for the same few counters the memory channel that owns those words took 3e10 loads per second, the update tiles that shared it ran 6 000 times
// slower -- 0.68 s for a 100 us tile, profiles/r06_chain_poll_storm.txt -- and the waits ran into their time limit: TLPK_INTERNAL in 15 % of the runs.)
// The time limit is wall-clock time (100 MHz counter), `timeout_ms` milliseconds (TLPK_CHAIN_TIMEOUT_MS, default 2000).
__device__ __forceinline__ bool chain_wait(const ChainItem &it, const unsigned *cnt, int *info, const int lane, const unsigned timeout_ms, const int dbg_slot) {
    const int n01 = it.n0 + it.n1, ntot = n01 + (it.w2 >= 0 ? 1 : 0);
    if (ntot == 0) return __hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    const bool act = lane < ntot;
    const int e = min(lane, ntot - 1);
    const int idx = (e < it.n0) ? it.w0 + e : ((e < n01) ? it.w1 + (e - it.n0) : it.w2);
    const unsigned need = (unsigned)((e < it.n0) ? it.need0 : ((e < n01) ? it.need1 : it.need2));
    const unsigned *p = cnt + idx;
    bool dead = false;
    unsigned spins = 0;
    const unsigned long long t0 = wall_clock64();
    unsigned v = need;
    if (act) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (!dead && !__all(v >= need)) {
        if (spins < 64u) __builtin_amdgcn_s_sleep(8);
        else if (spins < 512u) __builtin_amdgcn_s_sleep(32);
        else __builtin_amdgcn_s_sleep(127);
        if ((++spins & 63u) == 0u) {
            int flag = 0;
            if (lane == 0) flag = __hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_readfirstlane(flag) != 0) dead = true;
            else if (wall_clock64() - t0 > (unsigned long long)timeout_ms * 100000ull) {
                // (diagnostics: who gave up on what -- words 4.. of the status block, printed by the host with TLPK_CHAIN_DEBUG=1)
                if (v < need && __hip_atomic_exchange(info + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    info[4] = dbg_slot; info[5] = it.role; info[6] = idx; info[7] = (int)need; info[8] = (int)v; info[9] = (int)blockIdx.x; info[10] = (int)gridDim.x; info[11] = it.task;
                }
                if (lane == 0) __hip_atomic_store(info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
            }
        }
        if (act) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE acquire after the match: plain loads of the producers' data from here on
    return !dead;
}


// Arguments of a non-kernel function arrive in VECTOR registers, uniform or not: without help every address computed from them is vector arithmetic and every
// descriptor load a vector load (the stand-alone kernels get theirs from the kernel-argument segment, in scalar registers).  uni() / uni_ctx() tell the compiler
// what it cannot see across the call: these values are wave-uniform.
template <class T> __device__ __forceinline__ T *uni(T *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    // (through a GLOBAL-address-space pointer: a generic pointer that crossed a call is reached with flat_load / flat_store, which also tie up the LDS counter)
    typedef __attribute__((address_space(1))) T gT;
    return (T *)(gT *)(((unsigned long long)hi << 32) | lo);
}
// The LDS block travels as an LDS-address-space pointer (32 bits): as a generic pointer the callee would reach it with FLAT instructions -- behind the call the
// compiler no longer knows that it is LDS --; the cast back to a generic pointer happens inside the callee, where address-space inference sees its origin.
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ double *uni_lds(lds_double *p) {
    const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(lds_char *)p);
    return (double *)(lds_double *)(lds_char *)(unsigned long long)v;
}
__device__ __forceinline__ DevCtx uni_ctx(const DevCtx &c) {
    DevCtx u = c;
    u.fronts = uni(c.fronts); u.Lval = uni(c.Lval); u.U0 = uni(c.U0); u.U1 = uni(c.U1); u.dinv = uni(c.dinv); u.spart = uni(c.spart); u.info = uni(c.info);
    u.csign = uni(c.csign); u.upd_seg = uni(c.upd_seg);
    return u;
}

template <bool SIGNED>
__device__ __noinline__ void chain_role_update(const UpdateTask *tp_, const DevCtx &c_, lds_double *lds_) {
    const UpdateTask *tp = uni(tp_); const DevCtx c = uni_ctx(c_); double *lds = uni_lds(lds_);
    const UpdateTask t{tp->front, tp->k0, tp->kw, tp->i0, tp->j0, tp->jlim, tp->beta0, tp->pad1, tp->seg, tp->nsl, tp->pad2, 0};
    const FrontDesc fd = c.fronts[t.front];
    if (t.pad2 == 2) { update_tile32<SIGNED>(t, fd, c); return; }         // (workgroup-uniform) a 32 x 32 tile of the next diagonal block: one 16 x 16 block per wave
    if (t.pad2) { update_tile64<SIGNED>(t, fd, c, lds); return; }          // a 64 x 64 tile (TLPK_CHAIN_TILE64=1)
    double (*As)[UPD_KT * UPD_LD] = reinterpret_cast<double (*)[UPD_KT * UPD_LD]>(lds);
    const bool full = (t.i0 + TILE <= fd.f) && (t.j0 + TILE <= t.jlim) && (t.i0 >= t.j0 + TILE);
    if (full) update_tile<true, SIGNED, 4>(t, fd, c, As, As + 2);
    else update_tile<false, SIGNED, 4>(t, fd, c, As, As + 2);
}
template <bool SIGNED>
__device__ __noinline__ void chain_role_potrf(const PotrfTask *tp_, const DevCtx &c_, lds_double *lds_, unsigned *prog_) {
    const PotrfTask *tp = uni(tp_); const DevCtx c = uni_ctx(c_); double *lds = uni_lds(lds_);
    const PotrfTask t{tp->front, tp->k0, tp->nb, tp->kprev};
    const FrontDesc fd = c.fronts[t.front];
    potrf_wide_task<SIGNED, 3>(t, fd, c, lds, prog_ ? uni(prog_) : nullptr);       // (prog_: the block column's counter when its strips enter early, see trsm_task_dma)
    __builtin_amdgcn_s_setprio(0);
}
template <bool SIGNED>
__device__ __noinline__ void chain_role_trsm(const TrsmTask *tp_, const DevCtx &c_, lds_double *lds_, lds_double *dyn_, unsigned *cnt_, const unsigned timeout_ms_) {
    const TrsmTask *tp = uni(tp_); const DevCtx c = uni_ctx(c_); double *lds = uni_lds(lds_);
    const TrsmTask t{tp->front, tp->k0, tp->nb, tp->row0, tp->kprev, tp->fuse_nb, tp->pad1, tp->pad2};
    const FrontDesc fd = c.fronts[t.front];
    if (t.nb == NB_OUT) {
        if constexpr (TRSM_DMA) {
            // (dyn_: the launch's dynamic LDS when it holds two more 64 x 64 images -- the default, one workgroup per CU --, else null: the ring of two;
            //  pad2 > 0: early entry, the counter of the block column's diagonal block is cnt[pad2 - 1], see trsm_task_dma)
            const TrsmProg pg{t.pad2 > 0 ? uni(cnt_) + (t.pad2 - 1) : nullptr, c.info, (unsigned)__builtin_amdgcn_readfirstlane((int)timeout_ms_)};
            if (dyn_) {
                if (t.pad2 > 0) trsm_task_dma<SIGNED, 4, true>(t, fd, c, lds, uni_lds(dyn_), pg);
                else trsm_task_dma<SIGNED, 4, false>(t, fd, c, lds, uni_lds(dyn_), pg);
            } else {
                if (t.pad2 > 0) trsm_task_dma<SIGNED, 2, true>(t, fd, c, lds, nullptr, pg);
                else trsm_task_dma<SIGNED, 2, false>(t, fd, c, lds, nullptr, pg);
            }
        } else trsm_task<SIGNED, true>(t, fd, c, reinterpret_cast<double (*)[NB_IN * LDW]>(lds));
    } else trsm_task<SIGNED, false>(t, fd, c, reinterpret_cast<double (*)[NB_IN * LDW]>(lds));
}
__device__ __noinline__ void chain_role_reduce(const UpdateTask *tp_, const int sub_, const DevCtx &c_) {
    const UpdateTask *tp = uni(tp_); const DevCtx c = uni_ctx(c_); const int sub = __builtin_amdgcn_readfirstlane(sub_);
    const UpdateTask t{tp->front, tp->k0, tp->kw, tp->i0, tp->j0, tp->jlim, tp->beta0, tp->pad1, tp->seg, tp->nsl, 0, 0};
    const FrontDesc fd = c.fronts[t.front];
    reduce_part(t, sub, fd, c);
}

template <bool SIGNED>
__global__ __launch_bounds__(256, 2) void k_chain(const ChainArgs a, DevCtx c) {
    __shared__ __attribute__((aligned(16))) double lds[CHAIN_LDS];
    extern __shared__ __attribute__((aligned(16))) double chain_dyn[];
    lds_double *dyn = a.dyn_lds >= 2u * TRSM_DMA_BLK * sizeof(double) ? (lds_double *)chain_dyn : (lds_double *)nullptr;
    unsigned *ctl = reinterpret_cast<unsigned *>(lds);        // ctl[0] = the drawn ticket, ctl[1] = 1: skip the work (somebody gave up waiting); in the role's own LDS, between its uses
    const int tid = threadIdx.x;
    for (;;) {
        __syncthreads();                                       // the previous item no longer reads its LDS
        if (tid == 0) ctl[0] = __hip_atomic_fetch_add(a.cnt + a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane((int)ctl[0]);
        if (slot >= (unsigned)a.nitems) return;
        const ChainItem *ip = a.items + slot;
        const ChainItem it{ip->role, ip->task, ip->sub, ip->w0, ip->n0, ip->need0, ip->w1, ip->n1, ip->need1, ip->w2, ip->need2, ip->sig};
        unsigned long long *tr = a.trace ? a.trace + 4 * (size_t)slot : nullptr;
        if (tr && tid == 0) tr[0] = wall_clock64();
        if (tid < 64) {
            const bool ok = chain_wait(it, a.cnt, c.info, tid, a.spin_max, (int)slot);
            if (tid == 0) ctl[1] = ok ? 0u : 1u;
        }
        __syncthreads();
        const bool skip = __builtin_amdgcn_readfirstlane((int)ctl[1]) != 0;
        __syncthreads();                                       // ctl is read: the role may overwrite it
        if (tr && tid == 0) tr[1] = wall_clock64();
        if (!skip) {
            // (the roles are separate functions, not inlined: each gets the register allocation of the stand-alone kernel it comes from -- inlined into
            // one loop body the strips' 256-register working set pushed 125 registers of the other roles' live ranges into scratch)
            if (it.role == CR_UPDATE) chain_role_update<SIGNED>(a.upd + it.task, c, (lds_double *)lds);
            else if (it.role == CR_POTRF) chain_role_potrf<SIGNED>(a.potrf + it.task, c, (lds_double *)lds, (it.sub && it.sig >= 0) ? a.cnt + it.sig : nullptr);
            else if (it.role == CR_TRSM) chain_role_trsm<SIGNED>(a.trsm + it.task, c, (lds_double *)lds, dyn, a.cnt, a.spin_max);
            else chain_role_reduce(a.red + it.task, it.sub, c);
        }
        // publish: every wave's stores (and L2 adds) have left the CU, then ONE lane writes the die's L2 back and raises the counter
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tr && tid == 0) tr[2] = wall_clock64();
        if (tid == 0 && it.sig >= 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(a.cnt + it.sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tr && tid == 0) tr[3] = wall_clock64();
    }
}

// ------------------------------------------------------------------------------------------
// solve kernels
// ------------------------------------------------------------------------------------------
// xw[ii] = xi_p[i] + sum_j A[i,j] D_j xi_d[j],  i = perm[ii].  Sharded runs: a rank sums only its
// own columns and only rank 0 adds xi_p on linking rows (the all-reduce completes the sum).
// w = D .* xi_d on this rank's columns (0 elsewhere), once per solve: the row kernel below then gathers ONE vector per
// entry of A instead of three (column mask, D, xi_d); the products are formed in the same order as before
__global__ void k_rhs_scale(i64 n, const double *__restrict__ D, const double *__restrict__ xi_d, const char *__restrict__ col_local, double *__restrict__ w,
                            const double *__restrict__ xi_d1) {
    if (blockIdx.y) { xi_d = xi_d1; w += n; }                     // (grid y = right-hand side of a pair)
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) w[j] = col_local[j] ? D[j] * xi_d[j] : 0.0;
}
__global__ __launch_bounds__(256) void k_rhs(i64 m, const i32 *__restrict__ perm, const i64 *__restrict__ Tp,
                      const i32 *__restrict__ Tj, const double *__restrict__ Tx,
                      const double *__restrict__ D, const double *__restrict__ xi_p,
                      const double *__restrict__ xi_d, const char *__restrict__ row_local,
                      const char *__restrict__ col_local, int rank, double *__restrict__ xw,
                      const double *__restrict__ xi_p1, i64 w2, i64 xw2) {
    if (blockIdx.y) { xi_p = xi_p1; D += w2; xw += xw2; }          // (grid y = right-hand side of a pair: its xi_p, pre-scaled vector and xw)
    // 8 lanes per row (LP rows are short: a wave per row left 7/8 of the lanes idle; a long linking
    // row just takes more trips); fixed shuffle-tree reduction inside the 8-lane group
    // Tp / Tj / Tx: the CSR copy with rows in PERMUTED order (tlpk_api.cpp: upload_all) -- row ii is contiguous with row ii + 1
    const i64 ii = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int lane = threadIdx.x & 7;
    const bool live = ii < m;
    const i64 q0 = live ? Tp[ii] : 0, q1 = live ? Tp[ii + 1] : 0;
    const i32 i = live ? perm[ii] : 0;
    const char rl = live ? row_local[i] : 0;
    double s = 0.0;
    if (rl != 0) {
        for (i64 q = q0 + lane; q < q1; q += 8) s += Tx[q] * D[Tj[q]];      // D = the pre-scaled vector w (k_rhs_scale)
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) s += __shfl_down(s, off, 8);
    if (live && lane == 0) xw[ii] = (rl == 0) ? 0.0 : (((rl == 2 && rank != 0) ? 0.0 : xi_p[i]) + s);
}

// forward gather: row t of the front receives the entries of its children's contribution vectors
// listed in gth_src (child order = summation order): pivot rows add them to xw, rows below the
// pivot block start the front's own contribution vector uc.  One thread per row, no conflicts.
__global__ __launch_bounds__(256) void k_fwd_gather(const SolveTask *__restrict__ tasks, DevCtx c) {
    if (blockIdx.y) { c.xw += c.xw2; c.uc += c.uc2; }      // second right-hand side of a pair: its copies of xw / uc (one launch for both)
    const SolveTask t = tasks[blockIdx.x];
    const FrontDesc fd = c.fronts[t.front];
    if (t.nb == SOLVE_ROWS / 8) {
        // wide fan-in (the root front of a block-angular LP: every linking row collects one entry per diagonal block): the
        // task holds 32 rows, 8 lanes per row, fixed shuffle tree -- one thread per row walked 64 dependent loads (53 us
        // for the 1000 rows of the root in 4 workgroups, on every solve)
        const int l8 = threadIdx.x & 7;
        const i32 rr = t.row0 + (threadIdx.x >> 3);
        const bool live = rr < fd.f;
        const i64 a0 = live ? c.gth_ptr[fd.rowoff + rr] : 0, a1 = live ? c.gth_ptr[fd.rowoff + rr + 1] : 0;
        double s = 0.0;
        for (i64 q = a0 + l8; q < a1; q += 8) s += c.uc[c.gth_src[q]];
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) s += __shfl_down(s, off, 8);
        if (live && l8 == 0 && !(rr < fd.ns && a0 == a1)) {
            double *d = (rr < fd.ns) ? (c.xw + fd.col0 + rr) : (c.uc + fd.ucoff + (rr - fd.ns));
            *d = ((rr < fd.ns) ? *d : 0.0) + s;
        }
        return;
    }
    const i32 r = t.row0 + threadIdx.x;
    if (r >= fd.f) return;
    const i64 q0 = c.gth_ptr[fd.rowoff + r], q1 = c.gth_ptr[fd.rowoff + r + 1];
    double *dst = (r < fd.ns) ? (c.xw + fd.col0 + r) : (c.uc + fd.ucoff + (r - fd.ns));
    if (r < fd.ns && q0 == q1) return;
    double v = (r < fd.ns) ? *dst : 0.0;
    for (i64 q = q0; q < q1; ++q) v += c.uc[c.gth_src[q]];
    *dst = v;
}

// ------------------------------------------------------------------------------------------
// Hand-over of a solved block between workgroups of ONE launch (persistent sweeps below).
// /opt/skills/guides/cdna_hip_programming.md Guideline 16, the "8-byte agent-scope atomics on both
// sides" form: the payload (<= 128 doubles) is stored with relaxed agent-scope atomic stores
// (write-through `sc1` stores), every storing wave drains its stores (s_waitcnt vmcnt(0)), a
// barrier, then ONE lane stores the flag (relaxed, agent scope).  The consumer polls the flag with
// relaxed agent-scope loads from one lane (s_sleep between polls), a barrier, then reads the
// payload with relaxed agent-scope atomic loads (`sc1`: never served from the reading CU's L1).
// No fences, nothing depends on placement.  Round 1's attempt failed exactly where the guide says it
// must: plain payload stores + relaxed flag (stale half the time), or __threadfence() per workgroup.
// Every spin is bounded: a workgroup that waits longer than ~2 s sets info[1] and everybody leaves
// (the host reports TLPK_INTERNAL) -- a scheduling bug must never hang the device.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double ld_agent(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane: wait until *flag == epoch; false = aborted (timeout here or elsewhere)
__device__ __forceinline__ bool wait_flag(const unsigned *flag, unsigned epoch, int *info) {
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(4);
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spins > (1u << 21)) { __hip_atomic_store(info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
    }
    asm volatile("" ::: "memory");        // nothing that follows may be scheduled above the poll
    return true;
}

// ------------------------------------------------------------------------------------------
// Diagonal blocks of the triangular solves (nb <= SOLVE_NB = 2 sub-blocks of NB_IN) with the
// inverted 64 x 64 sub-blocks written by k_potrf*:
//   forward   y1 = Wa b1 ;  y2 = Wb (b2 - L21 y1)
//   backward  x2 = Wb' t2 ;  x1 = Wa' (t1 - L21' x2)
// A 64-long dot product is split over the 4 waves (thread (i, part) takes terms k = 16 part ..
// 16 part + 15 of row/column i), partials combined in fixed order.  The diagonal block sits on the
// sweep's serial chain (one block per launch), so the operand fragments -- 3 x 16 values per
// thread -- are fetched into registers at the START of the kernel, while the bulk part of the
// launch streams its panel rows; addresses are clamped and out-of-range terms zeroed by a select
// (no per-lane branches around the loads).
// ------------------------------------------------------------------------------------------
constexpr int FWD_DIAG_SCRATCH = 2 * SOLVE_NB + 4 * NB_IN;     // doubles of LDS: rhs, result, partials

// One operand fragment (16 values per thread) of a diagonal-block product.  WHICH: 0 = Wa (inverse of
// the first 64 x 64 sub-block), 1 = L21, 2 = Wb (inverse of the second sub-block).
//   forward : thread (i = tid & 63, part = tid >> 6) gets M[i][16 part + kk]        (lanes along rows)
//   backward: thread (lane = k, part) gets M[k][ci], ci = 16 part + kk               (lanes along rows)
// both coalesced.  Addresses are clamped and out-of-range terms zeroed by a select afterwards: the
// 16 loads are unconditional straight-line code (a branch around each load makes every load wait
// for the previous one -- that was 22 us of a 37 us diagonal workgroup).
template <bool BACKWARD, int WHICH>
__device__ __forceinline__ void load_frag(const DevCtx &c, const FrontDesc &fd, const i32 bk0, const i32 nb, double (&w)[16]) {
    const i32 na = min(nb, NB_IN), nb2 = nb - na;
    const i32 lda = fd.lda;                         // leading dimension of the panel (>= f, multiple of 16 for large fronts)
    const int i = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double *M; i64 ld; i32 nr, nc;                   // matrix, leading dimension, rows, columns
    if (WHICH == 0) { M = front_dinv(c, fd, bk0); ld = na; nr = na; nc = na; }
    else if (WHICH == 1) { M = pcol(c, fd, bk0) + (bk0 + NB_IN); ld = lda - bk0; nr = nb2; nc = na; }       // packed panel: the slice of column bk0 (a multiple of 64)
    else { M = front_dinv(c, fd, bk0 + NB_IN); ld = nb2; nr = nb2; nc = nb2; }
    const i32 ir = min(i, nr - 1);
    double v[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) v[kk] = M[(i64)ir + (i64)min(16 * part + kk, nc - 1) * ld];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const i32 col = 16 * part + kk;
        bool ok = i < nr && col < nc;
        if (WHICH != 1) ok = ok && (i >= col);             // the inverses are lower triangular
        w[kk] = ok ? v[kk] : 0.0;
    }
}

// sum over the thread's 16 terms, then over the 4 parts in fixed order; result valid where part == 0
__device__ __forceinline__ double dot4(const double (&w)[16], const double *v, int i, int part, double (*ps)[NB_IN]) {
    double s = 0.0;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) s += w[kk] * v[16 * part + kk];
    ps[part][i] = s;
    __syncthreads();
    const double r = ((ps[0][i] + ps[1][i]) + ps[2][i]) + ps[3][i];
    __syncthreads();
    return r;
}

// scratch layout: vin[SOLVE_NB] (rhs of the block, zero beyond nb, filled by the caller, then a
// barrier) | vout[SOLVE_NB] | ps[4][NB_IN].  The result is written to xw[col0 + bk0 ..).
// PUBLISH: the result is stored with agent-scope (write-through) stores for a hand-over inside the launch.
// The three operand fragments are requested before the first product: one memory round trip instead of three
// on the sweep's serial chain.
template <bool PUBLISH = false>
__device__ __forceinline__ void fwd_diag_solve(const DevCtx &c, const FrontDesc &fd, const i32 bk0, const i32 nb, double *scratch) {
    double *bs = scratch, *ys = scratch + SOLVE_NB;
    double (*ps)[NB_IN] = reinterpret_cast<double (*)[NB_IN]>(scratch + 2 * SOLVE_NB);
    const i32 na = min(nb, NB_IN), nb2 = nb - na;
    const int tid = threadIdx.x, i = tid & 63, part = tid >> 6;
    double w[16], w1[16], w2[16];
    load_frag<false, 0>(c, fd, bk0, nb, w);
    if (nb2 > 0) {                                             // workgroup-uniform
        load_frag<false, 1>(c, fd, bk0, nb, w1);
        load_frag<false, 2>(c, fd, bk0, nb, w2);
    }
    const double y = dot4(w, bs, i, part, ps);                 // y1 = Wa b1
    if (part == 0) ys[i] = y;                                  // zero for i >= na (masked fragment)
    __syncthreads();
    if (nb2 > 0) {
        const double sl = dot4(w1, ys, i, part, ps);           // L21 y1
        if (part == 0) bs[NB_IN + i] -= sl;
        __syncthreads();
        const double y2 = dot4(w2, bs + NB_IN, i, part, ps);   // y2 = Wb (b2 - L21 y1)
        if (part == 0) ys[NB_IN + i] = y2;
        __syncthreads();
    }
    if (tid < nb) {
        if (PUBLISH) st_agent(c.xw + fd.col0 + bk0 + tid, ys[tid]);
        else c.xw[fd.col0 + bk0 + tid] = ys[tid];
    }
}

// column sums over the 64 lanes (rows k) of 16 products per lane; lane 0 ends up with the 16 sums
__device__ __forceinline__ void reduce16(double (&p)[16]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) p[kk] += __shfl_down(p[kk], off);
    }
}

// scratch layout: t[SOLVE_NB] (rhs, zero beyond nb) | x[SOLVE_NB]
template <bool PUBLISH = false>
__device__ __forceinline__ void bwd_diag_solve(const DevCtx &c, const FrontDesc &fd, const i32 bk0, const i32 nb, double *scratch) {
    double *ts = scratch, *xo = scratch + SOLVE_NB;
    const i32 na = min(nb, NB_IN), nb2 = nb - na;
    const int tid = threadIdx.x, lane = tid & 63, part = tid >> 6;
    double w[16], w1[16], w2[16];
    load_frag<true, 0>(c, fd, bk0, nb, w);                         // all fragments requested up front
    if (nb2 > 0) {
        load_frag<true, 2>(c, fd, bk0, nb, w2);
        load_frag<true, 1>(c, fd, bk0, nb, w1);
        const double t2 = ts[NB_IN + lane];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) w2[kk] *= t2;              // x2[ci] = sum_k Wb[k][ci] t2[k]
        reduce16(w2);
        if (lane == 0) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) xo[NB_IN + 16 * part + kk] = w2[kk];     // zero for ci >= nb2
        }
        __syncthreads();
        const double x2 = xo[NB_IN + lane];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) w1[kk] *= x2;              // sum_k L21[k][ci] x2[k]
        reduce16(w1);
        if (lane == 0) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) ts[16 * part + kk] -= w1[kk];
        }
        __syncthreads();
    }
    const double t1 = ts[lane];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) w[kk] *= t1;                   // x1[ci] = sum_k Wa[k][ci] t1[k]
    reduce16(w);
    if (lane == 0) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) xo[16 * part + kk] = w[kk];
    }
    __syncthreads();
    if (tid < nb) {
        if (PUBLISH) st_agent(c.xw + fd.col0 + bk0 + tid, xo[tid]);
        else c.xw[fd.col0 + bk0 + tid] = xo[tid];
    }
}

// first block of every front of a level: nothing to overlap with
__global__ __launch_bounds__(256) void k_fwd_diag(const SolveTask *__restrict__ tasks, DevCtx c) {
    __shared__ double scratch[FWD_DIAG_SCRATCH];
    const SolveTask t = tasks[blockIdx.x];
    const FrontDesc fd = c.fronts[t.front];
    if (threadIdx.x < SOLVE_NB) scratch[threadIdx.x] = (threadIdx.x < t.nb) ? c.xw[fd.col0 + t.k0 + threadIdx.x] : 0.0;
    __syncthreads();
    fwd_diag_solve(c, fd, t.k0, t.nb, scratch);
}

// forward update: rows below a solved block: rhs[r] -= sum_j L[r, k0+j] * y[j].  The workgroup
// that owns the first SOLVE_ROWS rows below the block also holds the NEXT block's rows: when
// t.nslot (= width of the next block) is set it solves that diagonal block right away
// (look-ahead), so the forward sweep needs one launch per block instead of two; the new rhs of
// that block goes from the update straight into LDS.
__global__ __launch_bounds__(256) void k_fwd_update(const SolveTask *__restrict__ tasks, DevCtx c) {
    __shared__ double ys[SOLVE_NB];
    __shared__ double scratch[FWD_DIAG_SCRATCH];
    const SolveTask t = tasks[blockIdx.x];
    const FrontDesc fd = c.fronts[t.front];
    const i32 f = fd.f, ns = fd.ns, nb = t.nb;
    const i32 lda = fd.lda;                         // leading dimension of the panel (>= f, multiple of 16 for large fronts)
    const bool fused = t.nslot > 0;                      // workgroup-uniform
    const double *P = c.Lval + fd.loff;             // packed panel: column k at P + pk_off(lda, k)
    const i32 r = t.row0 + threadIdx.x, rc = min(r, f - 1);
    // the first panel columns travel while the solved block is staged
    double pv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) pv[j] = P[(i64)rc + pk_off(lda, t.k0 + min(j, nb - 1))];
    if (threadIdx.x < SOLVE_NB) ys[threadIdx.x] = (threadIdx.x < nb) ? c.xw[fd.col0 + t.k0 + threadIdx.x] : 0.0;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += pv[j] * ys[j];             // ys is zero beyond nb
    // 16 columns per trip, all 16 loads in flight (the compiler unrolled the plain loop by 4: 28
    // dependent round trips per block step on the sweep's serial chain); columns clamped, ys zero
    // beyond nb
    for (i32 jb = 16; jb < nb; jb += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) pv[j] = P[(i64)rc + pk_off(lda, t.k0 + min(jb + j, nb - 1))];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += pv[j] * ys[jb + j];
    }
    double v = 0.0;
    if (r < f) {
        double *dst = (r < ns) ? (c.xw + fd.col0 + r) : (c.uc + fd.ucoff + (r - ns));
        v = *dst - acc;
        *dst = v;
    }
    if (fused) {
        // rows row0 .. row0 + nslot - 1 are the next diagonal block (row0 == k0 + nb, all pivot rows)
        if (threadIdx.x < SOLVE_NB) scratch[threadIdx.x] = (threadIdx.x < t.nslot) ? v : 0.0;
        __syncthreads();
        fwd_diag_solve(c, fd, t.k0 + nb, t.nslot, scratch);
    }
}

// backward step (column-oriented, the mirror image of the forward sweep): the rows
// [row0, row0 + nrows) of the front hold final solution values (the front's rows below the pivot
// block come from the ancestors, a pivot block from an earlier launch); their contribution is
// removed from the rhs of ONE earlier column block:  t[k0 + j] -= sum_r L[r, k0 + j] * x[r].
// Every column block is owned by exactly one workgroup per launch (no partial-sum slots, no
// atomics, fixed summation order).  The workgroup of the column block right above the source rows
// (t.nslot != 0) has then seen every contribution to its block and solves the diagonal block at
// once, so the backward sweep needs one launch per block.
// A wave takes 8 columns at a time (16 independent 512-byte loads in flight, the next 8 columns
// are fetched while the current ones are reduced), lanes run along the contiguous rows;
// shuffle-tree reduction.
__global__ __launch_bounds__(256, 4) void k_bwd_update(const SolveTask *__restrict__ tasks, DevCtx c) {
    __shared__ double scratch[FWD_DIAG_SCRATCH];
    __shared__ double tacc[SOLVE_NB];       // running sums per column (a column belongs to one wave)
    const SolveTask t = tasks[blockIdx.x];
    const FrontDesc fd = c.fronts[t.front];
    const i32 ns = fd.ns, nb = t.nb, nrows = t.slot;
    const i32 lda = fd.lda;                         // leading dimension of the panel (>= f, multiple of 16 for large fronts)
    const bool fused = t.nslot != 0;                     // workgroup-uniform
    const i32 *rows = c.rowidx + fd.rowoff;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < SOLVE_NB) tacc[threadIdx.x] = 0.0;
    __syncthreads();
    const double *P0 = c.Lval + fd.loff;            // packed panel: column k at P0 + pk_off(lda, k)
    for (i32 rc = t.row0; rc < t.row0 + nrows; rc += BWD_ROWS) {
        const i32 nr = min(BWD_ROWS, t.row0 + nrows - rc);
        double xr[BWD_ROWS / 64];
        i32 ro[BWD_ROWS / 64];
        i32 xi[BWD_ROWS / 64];
#pragma unroll
        for (int u = 0; u < BWD_ROWS / 64; ++u) {           // unconditional loads, clamped rows, selects
            ro[u] = min(lane + 64 * u, nr - 1);           // clamped: the matching xr is zeroed
            const i32 r = rc + ro[u];
            const i32 gr = rows[r];
            xi[u] = (r < ns) ? (fd.col0 + r) : gr;
        }
#pragma unroll
        for (int u = 0; u < BWD_ROWS / 64; ++u) {
            const double xv = c.xw[xi[u]];
            xr[u] = (lane + 64 * u < nr) ? xv : 0.0;
        }
        const double *P = P0 + rc;
        // a wave owns 32 of the 128 columns: batches of 8, the next batch is requested while the current
        // one is reduced (16-column batches need 212 registers and cost the streaming workgroups their
        // occupancy: 8.0 vs 7.3 ms per step)
        constexpr int CB = 8;
        double cur[CB][BWD_ROWS / 64], nxt[CB][BWD_ROWS / 64];
        auto fetch = [&](double (&dst)[CB][BWD_ROWS / 64], const i32 j0) {
#pragma unroll
            for (int jj = 0; jj < CB; ++jj) {
                const double *col = P + pk_off(lda, t.k0 + min(j0 + jj, nb - 1));      // clamped: extra columns are dropped below
#pragma unroll
                for (int u = 0; u < BWD_ROWS / 64; ++u) dst[jj][u] = col[ro[u]];
            }
        };
        fetch(cur, wave * CB);
#pragma unroll 1
        for (i32 j0 = wave * CB; j0 < nb; j0 += 4 * CB) {
            fetch(nxt, j0 + 4 * CB);                 // unconditional (columns are clamped): a guarded fetch
                                                     // forces a full wait at the join and hides nothing
            double acc[CB];
#pragma unroll
            for (int jj = 0; jj < CB; ++jj) {
                double a = 0.0;
#pragma unroll
                for (int u = 0; u < BWD_ROWS / 64; ++u) a += cur[jj][u] * xr[u];
                acc[jj] = a;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
                for (int jj = 0; jj < CB; ++jj) acc[jj] += __shfl_down(acc[jj], off);
            }
            if (lane == 0) {
#pragma unroll
                for (int jj = 0; jj < CB; ++jj) if (j0 + jj < nb) tacc[j0 + jj] += acc[jj];
            }
#pragma unroll
            for (int jj = 0; jj < CB; ++jj)
#pragma unroll
                for (int u = 0; u < BWD_ROWS / 64; ++u) cur[jj][u] = nxt[jj][u];
        }
    }
    __syncthreads();
    // the block's rhs after this launch's contributions: back to xw, and (diagonal workgroup) into LDS
    if (threadIdx.x < SOLVE_NB) {
        double v = 0.0;
        if (threadIdx.x < nb) {
            double *dst = c.xw + fd.col0 + t.k0 + threadIdx.x;
            v = *dst - tacc[threadIdx.x];
            if (nrows > 0 && !fused) *dst = v;
        }
        scratch[threadIdx.x] = v;
    }
    if (fused) {
        __syncthreads();
        bwd_diag_solve(c, fd, t.k0, nb, scratch);
    }
}

// ------------------------------------------------------------------------------------------
// Persistent sweeps: the whole forward (backward) substitution of every non-small front of a tree
// level in ONE launch.  The block steps of a front are a serial chain (56 steps of 64 columns for a
// 3500-column front, 760 for the 48 000-column front of the general sparse benchmark); with one launch
// per step the chain costs ~25 us per step (launch + a diagonal workgroup that starts cold + three
// dependent memory round trips in it), and every launch ends with a tail.  Here a workgroup owns a
// chunk of rows (forward) or a block of columns (backward) of one front for the whole sweep, keeps its
// partial sums in registers, streams its part of L exactly once, and receives each solved 64-wide
// block from the workgroup that solved it:
//   * hand-over words: one double per pivot column in a buffer that the host fills with a sentinel (all
//     ones, a NaN no computation produces) before every solve.  The producer stores its 64 results with
//     agent-scope atomic stores (write-through), a consumer WAVE polls the words it needs with agent-scope
//     atomic loads until none is the sentinel -- the data is its own flag (8-byte stores are not torn),
//     one memory round trip per hand-over, no fence, nothing placement-dependent
//     (/opt/skills/guides/cdna_hip_programming.md Guideline 16, forms R2 / "agent atomics both sides");
//   * waves stream independently (no workgroup barrier per block): a wave polls only the words of ITS
//     columns and broadcasts them from lane registers;
//   * L is static: the panel entries for the next two blocks are requested before the wait for the
//     current one, and the workgroup of a pivot block holds its inverted diagonal block in registers from
//     the start, so behind the last hand-over of a pivot block there are 16 multiply-adds, two LDS
//     reductions and the stores;
//   * items are handed out through a ticket counter in an order in which an item only waits for items
//     with SMALLER tickets (symbolic.cpp): those are held by workgroups that already run, so the
//     sweep cannot deadlock whatever the dispatch order or residency;
//   * fixed summation order per item => bitwise deterministic, independent of scheduling;
//   * every spin is bounded (~2 s): a wave that gives up sets info[1], everybody stops waiting, the host
//     reports TLPK_INTERNAL.  A scheduling bug must never hang the device.
// ------------------------------------------------------------------------------------------
constexpr unsigned long long SW_SENTINEL = 0xFFFFFFFFFFFFFFFFull;

// one wave: wait until the words xh[0 .. cnt) (cnt <= 64) hold results; lane l returns word min(l, cnt-1).
// Polls back off: the first `nfast` polls come every ~`fast` x 27 ns (a block that is about to be published: the
// chain of pivot blocks), later ones every ~`slow` x 27 ns (blocks that are many steps away: hundreds of waiting
// waves must not flood the L2 / fabric with coherent reads).
__device__ __forceinline__ double poll_block(const double *xh, const int cnt, const int lane, int *info, bool &dead, const SweepArgs &a) {
    const unsigned long long *p = reinterpret_cast<const unsigned long long *>(xh) + min(lane, cnt - 1);
    unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (!dead && !__all(v != SW_SENTINEL)) {
        const int naps = (spins < (unsigned)a.poll_nfast) ? a.poll_fast : a.poll_slow;
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(1);
        if ((++spins & 127u) == 0u) {
            if (__hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) dead = true;
            else if (spins > (1u << 21)) { __hip_atomic_store(info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; }
        }
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::: "memory");        // nothing that follows may be scheduled above the poll
    return __longlong_as_double((long long)v);
}
// two right-hand sides: the words of both must have arrived (the producer publishes them back to back)
__device__ __forceinline__ void poll_block2(const double *xh0, const double *xh1, const int cnt, const int lane, int *info, bool &dead,
                                            const SweepArgs &a, double &o0, double &o1) {
    const unsigned long long *p0 = reinterpret_cast<const unsigned long long *>(xh0) + min(lane, cnt - 1);
    const unsigned long long *p1 = reinterpret_cast<const unsigned long long *>(xh1) + min(lane, cnt - 1);
    unsigned long long v0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (!dead && !__all(v0 != SW_SENTINEL && v1 != SW_SENTINEL)) {
        const int naps = (spins < (unsigned)a.poll_nfast) ? a.poll_fast : a.poll_slow;
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(1);
        if ((++spins & 127u) == 0u) {
            if (__hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) dead = true;
            else if (spins > (1u << 21)) { __hip_atomic_store(info + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; }
        }
        v0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::: "memory");
    o0 = __longlong_as_double((long long)v0); o1 = __longlong_as_double((long long)v1);
}

// PIVOT item: rows [k0, k0 + nb) (nb <= 64) are a pivot block; lane = row, wave = quarter of the 64 columns of a
// consumed block.  BELOW item: <= 128 rows below the pivot block; waves 0/1 = rows 0..63 / 64..127 of the chunk
// for columns 0..31 of a block, waves 2/3 the same rows for columns 32..63.
// NR = 2: two right-hand sides share one pass over L (HSD's h-system and predictor are independent: HSD/step.jl:63,79): every
// panel entry loaded once feeds two accumulators; the second right-hand side lives at xw + c.xw2, uc + c.uc2, xh + a.xh2; each is
// summed in the order of the single-rhs kernel (bit-identical results).
template <bool PIVOT, int NR>
__device__ __forceinline__ void fwd_sweep_item(const SolveTask &t, const FrontDesc &fd, const DevCtx &c, const SweepArgs &a, double *scratch) {
    const double *xh = a.xh;
    constexpr int NBATCH = PIVOT ? 1 : 2;                      // batches of 16 columns per wave and block
    constexpr int WCOLS = 16 * NBATCH;                         // columns of a block handled by a wave
    const i32 f = fd.f, ns = fd.ns;
    const i32 lda = fd.lda;                         // leading dimension of the panel (>= f, multiple of 16 for large fronts)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rloc = PIVOT ? lane : lane + 64 * (wave & 1);    // row of the chunk
    const int cp0 = PIVOT ? 16 * wave : 32 * (wave >> 1);      // first column of the wave inside a block
    const char *Lb = reinterpret_cast<const char *>(c.Lval + fd.loff);
    const unsigned roff = (unsigned)min(t.k0 + rloc, f - 1) * 8u;     // clamped row: its result is never stored
    const double *xhf = xh + fd.col0;
    const i32 nin = t.nslot;
    double w[16];
    if (PIVOT) load_frag<false, 0>(c, fd, t.k0, t.nb, w);      // inverted diagonal block, held from the start
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.0;
    double b0[16], b1[16];
    bool dead = false;
    auto issue = [&](double (&b)[16], const i32 j, const int q) {          // block j (clamped), batch q
        const i32 c0 = min(j, nin - 1) * SWEEP_NB + cp0 + 16 * q;
#pragma unroll
        for (int u = 0; u < 16; ++u)                             // clamped column: x is zero beyond the block
            b[u] = *reinterpret_cast<const double *>(Lb + (size_t)pk_off(lda, min(c0 + u, ns - 1)) * 8u + roff);      // packed panel
    };
    auto consume = [&](const double (&b)[16], const double (&xv)[NR], const int q) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r] += b[u] * readlane_f64(xv[r], 16 * q + u);
    };
    auto wait_block = [&](const i32 j, double (&xv)[NR]) {      // this wave's columns of block j (zero beyond its width)
        const i32 cnt = min(WCOLS, min(SWEEP_NB, ns - j * SWEEP_NB) - cp0);
#pragma unroll
        for (int r = 0; r < NR; ++r) xv[r] = 0.0;
        if (cnt <= 0) return;
        if (NR == 1) xv[0] = poll_block(xhf + j * SWEEP_NB + cp0, cnt, lane, c.info, dead, a);
        else poll_block2(xhf + j * SWEEP_NB + cp0, xhf + a.xh2 + j * SWEEP_NB + cp0, cnt, lane, c.info, dead, a, xv[0], xv[NR - 1]);
#pragma unroll
        for (int r = 0; r < NR; ++r) xv[r] = (lane < cnt) ? xv[r] : 0.0;
    };
    if (nin > 0) {
        if (PIVOT) { issue(b0, 0, 0); issue(b1, 1, 0); }
        else { issue(b0, 0, 0); issue(b1, 0, 1); }
    }
    double xv[NR];
    if (PIVOT) {
        for (i32 j = 0; j < nin; j += 2) {
            wait_block(j, xv);
            consume(b0, xv, 0); issue(b0, j + 2, 0);
            if (j + 1 < nin) {                                  // wave-uniform
                wait_block(j + 1, xv);
                consume(b1, xv, 0); issue(b1, j + 3, 0);
            }
        }
    } else {
        for (i32 j = 0; j < nin; ++j) {
            wait_block(j, xv);
            consume(b0, xv, 0); issue(b0, j + 1, 0);
            consume(b1, xv, 1); issue(b1, j + 1, 1);
        }
    }
    double (*ps)[NB_IN] = reinterpret_cast<double (*)[NB_IN]>(scratch + 2 * SOLVE_NB);      // 4 x 64 doubles
    if (PIVOT) {
        double *bs = scratch;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            double *xwr = c.xw + (r ? c.xw2 : 0);
            if (r) __syncthreads();
            ps[wave][lane] = acc[r];
            __syncthreads();
            if (wave == 0) {
                const double total = ((ps[0][lane] + ps[1][lane]) + ps[2][lane]) + ps[3][lane];
                bs[lane] = (lane < t.nb) ? xwr[fd.col0 + t.k0 + min(lane, t.nb - 1)] - total : 0.0;
            }
            __syncthreads();
            const double y = dot4(w, bs, lane, wave, ps);           // y = W (b - sum); valid in wave 0
            if (wave == 0 && lane < t.nb) {
                xwr[fd.col0 + t.k0 + lane] = y;
                st_agent(const_cast<double *>(xhf) + (r ? a.xh2 : 0) + t.k0 + lane, y);     // hand-over: the data is its own flag
            }
        }
    } else {
        double *red = &ps[0][0];                                // 256 doubles: [column half][row]
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r) __syncthreads();
            if (wave >= 2) red[rloc] = acc[r];
            __syncthreads();
            if (wave < 2 && rloc < t.nb) {
                double *dst = c.uc + (r ? c.uc2 : 0) + fd.ucoff + (t.k0 + rloc - ns);
                *dst = *dst - (acc[r] + red[rloc]);
            }
        }
    }
}

template <int NSM> __device__ __forceinline__ void fwd_small_body(const FrontDesc &fd, const DevCtx &c, const int lane);      // (defined with k_fwd_small / k_bwd_small below)
template <int NSM> __device__ __forceinline__ void bwd_small_body(const FrontDesc &fd, const DevCtx &c, const int lane);
// WS (round 6): the launch also holds the level's SMALL fronts, as items with slot == 2 (k0 = a group of four tasks of a.small: one front per wave, the bodies of
// k_fwd_small / k_bwd_small).  Only for levels whose sweep is a few hundred items (symbolic.cpp: TLPK_SOLVE_MERGE) -- the latency-bound LPs, where a launch costs more than
// the fronts in it: one launch per level and direction less (25fv47 class: 19 -> 14 launches per solve).  The bandwidth-bound levels keep the two kernels: the small-front
// body needs 136 registers, the sweeps run four waves per SIMD on 113.
template <int NR, bool WS = false>
__global__ __launch_bounds__(256, WS ? (NR == 1 ? 3 : 2) : (NR == 1 ? 4 : 3)) void k_fwd_sweep(const SolveTask *__restrict__ tasks, DevCtx c, SweepArgs a) {
    __shared__ double scratch[FWD_DIAG_SCRATCH];
    __shared__ unsigned s_item;
    if (threadIdx.x == 0) s_item = (unsigned)(atomicAdd(a.ticket, 1ULL) + 1ULL);      // the counter starts at all ones (see SweepArgs): first ticket = 0
    __syncthreads();
    const SolveTask t = tasks[s_item];
    if (WS && t.slot == 2) {                                     // (workgroup-uniform) four small fronts, one per wave
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const SolveTask ts = a.small[(i64)t.k0 * 4 + wave];
        if (ts.front < 0) return;
        const FrontDesc fs = c.fronts[ts.front];
        const bool few = fs.ns <= 4 && !c.small_full;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            DevCtx cr = c;
            if (r) { cr.xw += c.xw2; cr.uc += c.uc2; }
            if (few) fwd_small_body<4>(fs, cr, lane); else fwd_small_body<SMALL_NS>(fs, cr, lane);
        }
        return;
    }
    const FrontDesc fd = c.fronts[t.front];
    if (t.slot) fwd_sweep_item<true, NR>(t, fd, c, a, scratch);         // workgroup-uniform
    else fwd_sweep_item<false, NR>(t, fd, c, a, scratch);
}

// Backward: item = column block [k0, k0 + nb) (nb <= 64) of a front.  t[k0 + j] = b - sum_r L[r, k0 + j] x[r] over the
// rows below the pivot block (values of the ancestors, known at launch) and over the LATER pivot blocks of the
// front, consumed as they are published (last block first); then x = W' t and the publish.  Lanes run along the 64
// contiguous rows of a tile, a wave owns 16 of the columns and keeps per-lane partial sums over ALL tiles (one
// shuffle reduction at the end).
template <int NR, bool WS = false>
__global__ __launch_bounds__(256, NR == 1 ? 3 : 2) void k_bwd_sweep(const SolveTask *__restrict__ tasks, DevCtx c, SweepArgs a) {
    __shared__ double red[NB_IN][NB_IN + 1];        // per-lane partial sums of the 64 columns, transposed reduction
    __shared__ double ts[NB_IN], bsh[NR][NB_IN];
    __shared__ double ps[4][NB_IN];
    __shared__ unsigned s_item;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_item = (unsigned)(atomicAdd(a.ticket, 1ULL) + 1ULL);
    __syncthreads();
    const SolveTask t = tasks[s_item];
    if (WS && t.nslot == -2) {                                   // (workgroup-uniform) four small fronts of the level, one per wave (see k_fwd_sweep)
        const SolveTask tsm = a.small[(i64)t.k0 * 4 + wave];
        if (tsm.front < 0) return;
        const FrontDesc fs = c.fronts[tsm.front];
        const bool few = fs.ns <= 4 && !c.small_full;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            DevCtx cr = c;
            if (r) { cr.xw += c.xw2; cr.uc += c.uc2; }
            if (few) bwd_small_body<4>(fs, cr, lane); else bwd_small_body<SMALL_NS>(fs, cr, lane);
        }
        return;
    }
    const FrontDesc fd = c.fronts[t.front];
    const i32 f = fd.f, ns = fd.ns, nb = t.nb;
    const i32 *rows = c.rowidx + fd.rowoff;
    const double *xhf = a.xh + fd.col0;
    const i32 nblk = (ns + SWEEP_NB - 1) / SWEEP_NB;
    const i32 nbelow = (t.slot > 0) ? (f + 63) / 64 - ns / 64 : 0;   // tiles below the pivot block, ENDING on multiples of 64 rows (line-aligned loads)
    const i32 ntiles = nbelow + t.nslot;
    // this wave's columns: wave-uniform bases (scalar registers) + 32-bit lane offsets
    const char *Pb = reinterpret_cast<const char *>(pcol(c, fd, t.k0));         // packed panel: the block's columns lie in the slice of t.k0
    const i32 ldk = pld(fd, t.k0);
    // Held from the start, off the chain: this block's right-hand side and the fragment of the inverted diagonal
    // block for x[ci] = sum_k W[k][ci] t[k]: thread (ci = lane, part = wave) keeps W[16 part + kk][ci] (W is stored
    // column-major: 16 consecutive doubles per lane)
    if (tid < NB_IN) {
#pragma unroll
        for (int r = 0; r < NR; ++r) bsh[r][tid] = (tid < nb) ? c.xw[(r ? c.xw2 : 0) + fd.col0 + t.k0 + min(tid, nb - 1)] : 0.0;
    }
    double w[16];
    {
        const double *W = front_dinv(c, fd, t.k0);              // nb x nb, column-major, ld = nb, upper part zero
        const i32 ci = min(lane, nb - 1);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const i32 k = 16 * wave + kk;
            const double v = W[(i64)min(k, nb - 1) + (i64)ci * nb];
            w[kk] = (lane < nb && k < nb && k >= lane) ? v : 0.0;
        }
    }
    double acc[NR][16];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[r][u] = 0.0;
    bool dead = false;
    auto tile_r0 = [&](const i32 q) { const i32 qc = min(q, ntiles - 1); return qc < nbelow ? (qc == 0 ? ns : (ns / 64 + qc) * 64) : (nblk - 1 - (qc - nbelow)) * SWEEP_NB; };
    auto tile_nr = [&](const i32 q) { const i32 qc = min(q, ntiles - 1); const i32 r0 = tile_r0(qc); return qc < nbelow ? min(f, (ns / 64 + qc + 1) * 64) - r0 : min(SWEEP_NB, ns - r0); };
    auto issue = [&](double (&b)[16], const i32 q) {           // tile q (clamped): 64 rows x this wave's 16 columns
        const i32 r0 = tile_r0(q), nr = tile_nr(q);
        const unsigned rb = (unsigned)(r0 + min(lane, nr - 1)) * 8u;            // clamped row: its x is zeroed
#pragma unroll
        for (int u = 0; u < 16; ++u)
            b[u] = *reinterpret_cast<const double *>(Pb + (size_t)min(16 * wave + u, nb - 1) * (size_t)ldk * 8u + rb);   // clamped column: dropped below
    };
    auto row_index = [&](const i32 q) -> i32 {                  // global index of this lane's row in a tile below the pivot block
        const i32 qc = min(q, max(nbelow - 1, 0));
        const i32 r0 = (qc == 0) ? ns : (ns / 64 + qc) * 64, nr = min(f, (ns / 64 + qc + 1) * 64) - r0;
        return (nbelow > 0) ? rows[r0 + min(lane, max(nr - 1, 0))] : 0;
    };
    auto consume = [&](const double (&b)[16], const double (&xr)[NR]) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r][u] += b[u] * xr[r];
    };
    double b0[16], b1[16];
    if (ntiles > 0) { issue(b0, 0); issue(b1, 1); }
    // rows below the pivot block: x of the ancestors through the row indices -- indices two tiles ahead, values one
    i32 gi0 = row_index(0), gi1 = row_index(1);
    double xn[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) xn[r] = (nbelow > 0) ? c.xw[(r ? c.xw2 : 0) + gi0] : 0.0;       // x of tile 0
    auto tile_x = [&](const i32 q, double (&xv)[NR]) {          // x[row of this lane] for tile q
        const i32 nr = tile_nr(q);
        if (q < nbelow) {
#pragma unroll
            for (int r = 0; r < NR; ++r) { xv[r] = xn[r]; xn[r] = c.xw[(r ? c.xw2 : 0) + gi1]; }   // requested one tile ago; next: tile q + 1
            gi1 = row_index(q + 2);
        } else {
            const i32 jb = nblk - 1 - (q - nbelow);
            if (NR == 1) xv[0] = poll_block(xhf + jb * SWEEP_NB, nr, lane, c.info, dead, a);
            else poll_block2(xhf + jb * SWEEP_NB, xhf + a.xh2 + jb * SWEEP_NB, nr, lane, c.info, dead, a, xv[0], xv[NR - 1]);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) xv[r] = (lane < nr) ? xv[r] : 0.0;
    };
    double xr[NR];
    for (i32 q = 0; q < ntiles; q += 2) {
        tile_x(q, xr);
        consume(b0, xr); issue(b0, q + 2);
        if (q + 1 < ntiles) {                                   // workgroup-uniform
            tile_x(q + 1, xr);
            consume(b1, xr); issue(b1, q + 3);
        }
    }
    (void)gi0;
    // column sums over the 64 lanes (rows) through LDS: thread (col = tid / 4, sub = tid % 4) adds 16 lanes' partial
    // sums, the four subs are combined by two quad shuffles -- fixed order, ~40 LDS accesses on the chain instead of
    // two 96-shuffle trees
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (r) __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; ++u) red[16 * wave + u][lane] = acc[r][u];
        __syncthreads();
        {
            const int col = tid >> 2, sub = tid & 3;
            double s_ = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) s_ += red[col][16 * sub + i];
            s_ += __shfl_xor(s_, 1);
            s_ += __shfl_xor(s_, 2);
            if (sub == 0) ts[col] = bsh[r][col] - s_;               // zero beyond nb (clamped duplicate columns are dropped by w)
        }
        __syncthreads();
        double xs_ = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) xs_ += w[kk] * ts[16 * wave + kk];
        ps[wave][lane] = xs_;
        __syncthreads();
        if (wave == 0 && lane < nb) {
            const double x = ((ps[0][lane] + ps[1][lane]) + ps[2][lane]) + ps[3][lane];
            c.xw[(r ? c.xw2 : 0) + fd.col0 + t.k0 + lane] = x;
            st_agent(const_cast<double *>(xhf) + (r ? a.xh2 : 0) + t.k0 + lane, x);   // hand-over: the data is its own flag
        }
    }
}

// ------------------------------------------------------------------------------------------
// Small fronts (ns <= SMALL_NS pivot columns, any number of rows below): one WAVE per front does a
// whole sweep step of the front -- the leaf levels hold most of the fronts, and a 256-thread
// workgroup per front in two or three kernels per level is nearly all fixed latency.  Four fronts per
// workgroup; the task list of a launch is padded with front = -1.  No shared memory, no barriers.
//   forward : y = L11^{-1} b (inverted block W, lane i = row i), then rows below: uc[r] -= L[r,:] y
//   backward: t = b - L21' x_below (lanes along rows, shuffle reduction), then x = W' t
// ------------------------------------------------------------------------------------------
constexpr int SMALL_RPL = SMALL_ROWS / 64;       // rows below per lane
// (round 4) The bodies are instantiated for (NSM pivot columns, RPL rows below per lane) = (4, 1), (4, 4), (16, 1), (16, 4) and chosen per front
// (wave-uniform): most small fronts of the inequality LPs have one or two pivot columns and a handful of rows, and the single (16, 4) body
// requested 64 + 32 clamped duplicate loads per lane for each of them -- 1.4 ms of a 7 ms solve on the north-star instance.
// (round 6) The rows below the pivot block go through the wave 64 at a time in a LOOP -- one trip for 95 % of the small fronts of the north-star LP -- instead of
// as up to four register sets loaded up front: the kernel needed 222 registers for the 5 % of tall fronts and ran two waves per SIMD for everybody
// (k_fwd_small + k_bwd_small: 0.84 of the 6.7 ms of a north-star solve at 1.4 - 1.8 TB/s, profiles/r06_pmc_summary.md).  Per row the sum runs over the
// columns in the same order as before: same bits.
template <int NSM>
__device__ __forceinline__ void fwd_small_body(const FrontDesc &fd, const DevCtx &c, const int lane) {
    const i32 f = fd.f, ns = fd.ns, rs = f - ns;
    const i32 lda = fd.lda;                         // leading dimension of the panel (>= f, multiple of 16 for large fronts)
    const double *__restrict__ W = front_dinv(c, fd, 0);          // ns x ns, column-major, ld = ns, upper part zero
    const double *__restrict__ P = c.Lval + fd.loff;
    double *xs = c.xw + fd.col0;
    double *__restrict__ uc = c.uc + fd.ucoff;
    // every load of the first trip is requested up front (clamped addresses, selects afterwards): the wave is
    // alone with its front, so each dependent round trip would be paid in full
    const i32 ic = min(lane, ns - 1);
    double wv[NSM], bv[NSM], lv[NSM];
#pragma unroll
    for (int k = 0; k < NSM; ++k) {
        const i32 kc = min(k, ns - 1);
        wv[k] = W[(i64)ic + (i64)kc * ns];
        bv[k] = xs[kc];
    }
    i32 rr = min(lane, max(rs - 1, 0));                            // row below, clamped (rs may be 0: stays inside the panel)
    double uv = (rs > 0) ? uc[rr] : 0.0;
#pragma unroll
    for (int k = 0; k < NSM; ++k) lv[k] = P[(i64)min(ns + rr, f - 1) + (i64)min(k, ns - 1) * lda];
    double y = 0.0;
#pragma unroll
    for (int k = 0; k < NSM; ++k) y += (k < ns && k <= lane) ? wv[k] * bv[k] : 0.0;
    if (lane < ns) xs[lane] = y;
    double yv[NSM];
#pragma unroll
    for (int k = 0; k < NSM; ++k) yv[k] = __shfl(y, k);        // only k < ns is used
    for (i32 r0 = 0;;) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NSM; ++k) acc += (k < ns) ? lv[k] * yv[k] : 0.0;
        if (r0 + lane < rs) uc[r0 + lane] = uv - acc;
        r0 += 64;
        if (r0 >= rs) break;                                       // (wave-uniform)
        rr = min(r0 + lane, rs - 1);
        uv = uc[rr];
#pragma unroll
        for (int k = 0; k < NSM; ++k) lv[k] = P[(i64)(ns + rr) + (i64)min(k, ns - 1) * lda];
    }
}
__global__ __launch_bounds__(256) void k_fwd_small(const SolveTask *__restrict__ tasks, DevCtx c) {
    if (blockIdx.y) { c.xw += c.xw2; c.uc += c.uc2; }      // second right-hand side of a pair: its copies of xw / uc (one launch for both)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform, and the compiler must know: task, front descriptor and every column base stay in scalar registers)
    const SolveTask t = tasks[(i64)blockIdx.x * 4 + wave];
    if (t.front < 0) return;
    const FrontDesc fd = c.fronts[t.front];
    const bool few = fd.ns <= 4 && !c.small_full;      // wave-uniform
    if (few) fwd_small_body<4>(fd, c, lane); else fwd_small_body<SMALL_NS>(fd, c, lane);
}

template <int NSM>
__device__ __forceinline__ void bwd_small_body(const FrontDesc &fd, const DevCtx &c, const int lane) {
    const i32 f = fd.f, ns = fd.ns, rs = f - ns;
    const i32 lda = fd.lda;                         // leading dimension of the panel (>= f, multiple of 16 for large fronts)
    const double *__restrict__ P = c.Lval + fd.loff;
    const double *__restrict__ W = front_dinv(c, fd, 0);
    const i32 *__restrict__ rows = c.rowidx + fd.rowoff;
    double *xs = c.xw + fd.col0;
    const i32 ic = min(lane, ns - 1);
    double wv[NSM], bv[NSM], lv[NSM], acc[NSM];
    i32 gi = rows[min(ns + lane, f - 1)];                              // rows below: values of the ancestors
#pragma unroll
    for (int k = 0; k < NSM; ++k) {
        const i32 kc = min(k, ns - 1);
        wv[k] = W[(i64)kc + (i64)ic * ns];                           // column `lane` of W
        bv[k] = xs[kc];
        acc[k] = 0.0;
    }
    // the rows below, 64 per trip (one trip for almost every small front; see fwd_small_body): a lane adds its rows in ascending order, as before
    for (i32 r0 = 0;;) {
        const i32 r = min(ns + r0 + lane, f - 1);
#pragma unroll
        for (int k = 0; k < NSM; ++k) lv[k] = P[(i64)r + (i64)min(k, ns - 1) * lda];
        const double xv = c.xw[gi];
        const double xr = (r0 + lane < rs) ? xv : 0.0;
#pragma unroll
        for (int k = 0; k < NSM; ++k) acc[k] += lv[k] * xr;
        r0 += 64;
        if (r0 >= rs) break;                                           // (wave-uniform)
        gi = rows[min(ns + r0 + lane, f - 1)];
    }
#pragma unroll
    for (int k = 0; k < NSM; ++k) acc[k] = (k < ns) ? acc[k] : 0.0;
    // (four columns at a time: sixteen shuffles in flight held 60 temporaries and the kernel at two waves per SIMD)
    double x = 0.0;                                                  // x[i] = sum_{k >= i} W[k][i] (b[k] - sum[k])
#pragma unroll
    for (int k0 = 0; k0 < NSM; k0 += 4) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int k = k0; k < k0 + 4; ++k) acc[k] += __shfl_down(acc[k], off);
        }
#pragma unroll
        for (int k = k0; k < k0 + 4; ++k) {
            const double tk = bv[k] - readlane_f64(acc[k], 0);        // (lane 0's sum through scalar registers)
            x += (k < ns && k >= lane) ? wv[k] * tk : 0.0;
        }
    }
    if (lane < ns) xs[lane] = x;
}
__global__ __launch_bounds__(256) void k_bwd_small(const SolveTask *__restrict__ tasks, DevCtx c) {
    if (blockIdx.y) { c.xw += c.xw2; c.uc += c.uc2; }      // second right-hand side of a pair: its copies of xw / uc (one launch for both)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (wave-uniform, and the compiler must know: task, front descriptor and every column base stay in scalar registers)
    const SolveTask t = tasks[(i64)blockIdx.x * 4 + wave];
    if (t.front < 0) return;
    const FrontDesc fd = c.fronts[t.front];
    const bool few = fd.ns <= 4 && !c.small_full;      // wave-uniform
    if (few) bwd_small_body<4>(fd, c, lane); else bwd_small_body<SMALL_NS>(fd, c, lane);
}

// dy_shared != nullptr (single-process multi-device mode): the rows this rank OWNS (its block rows; the linking rows
// on rank 0) are also written into the job-wide result vector, which may live on a peer device (P2P stores).
__global__ void k_unpermute(i64 m, const i32 *__restrict__ perm, const char *__restrict__ row_local,
                            const double *__restrict__ xw, double *__restrict__ dy, double *__restrict__ dy_shared, int rank,
                            double *__restrict__ dy1, i64 xw2) {
    if (blockIdx.y) { dy = dy1; xw += xw2; }                       // (grid y = right-hand side of a pair)
    const i64 ii = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (ii >= m) return;
    const i32 i = perm[ii];
    const char rl = row_local[i];
    const double v = rl ? xw[ii] : 0.0;
    dy[i] = v;
    if (dy_shared && (rl == 1 || (rl == 2 && rank == 0))) dy_shared[i] = v;
}

// dx_j = D_j (A[:,j]' dy - xi_d[j]);  columns of other ranks give 0 (local_only: are left alone -- dx is then the
// job-wide vector that every rank fills with its own columns).
__global__ void k_dx(i64 n, const i64 *__restrict__ Ap, const i32 *__restrict__ Ai,
                     const double *__restrict__ Ax, const double *__restrict__ D,
                     const double *__restrict__ dy, const double *__restrict__ xi_d,
                     const char *__restrict__ col_local, double *__restrict__ dx, int local_only,
                     const double *__restrict__ dy1, const double *__restrict__ xi_d1, double *__restrict__ dx1) {
    if (blockIdx.y) { dy = dy1; xi_d = xi_d1; dx = dx1; }          // (grid y = right-hand side of a pair)
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (!col_local[j]) { if (!local_only) dx[j] = 0.0; return; }
    double s = 0.0;
    for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) s += Ax[p] * dy[Ai[p]];
    dx[j] = D[j] * (s - xi_d[j]);
}

// Optional iterative refinement (tlpk_options.refine_steps; the reference leaves it as a TODO, spd.jl:68): residuals of the
// augmented system for the solution just computed,
//   r1 = xi_p - A dx - Rd dy   (rows, CSR)        r2 = xi_d + (theta + Rp) dx - A' dy   (columns, CSC)
// a second solve with (r1, r2) gives the correction that k_axpy2 adds.
__global__ void k_resid_rows(i64 m, const i64 *__restrict__ Tp, const i32 *__restrict__ Tj, const double *__restrict__ Tx,
                             const double *__restrict__ xi_p, const double *__restrict__ regD, const double *__restrict__ dx,
                             const double *__restrict__ dy, const char *__restrict__ row_local, int rank, int xip_all, double *__restrict__ r1) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double s = 0.0;
    for (i64 q = Tp[i]; q < Tp[i + 1]; ++q) s += Tx[q] * dx[Tj[q]];
    // sharded: dx is zero outside this rank's columns, so a linking row gets the rank's partial sum; the SUM over the ranks must be the row's
    // residual (the reduction of the root right-hand side inside the following solve, or the host's sum of the shards' values, completes it):
    //   xi_p: rank 0's only -- or, xip_all (device-resident loops: every shard holds its PARTIAL xi_p on the linking rows), every rank's;
    //   - Rd dy: dy is replicated on the linking rows, so it is counted ONCE, by rank 0, in either convention (round-5 advisor finding: with
    //   xip_all every shard subtracted it, a bias of (N - 1) Rd |dy| in the norms the refinement guard compares).
    const bool link = row_local[i] == 2;
    const double base = ((!link || rank == 0 || xip_all) ? xi_p[i] : 0.0) - ((!link || rank == 0) ? regD[i] * dy[i] : 0.0);
    r1[i] = base - s;
}
__global__ void k_resid_cols(i64 n, const i64 *__restrict__ Ap, const i32 *__restrict__ Ai, const double *__restrict__ Ax,
                             const double *__restrict__ xi_d, const double *__restrict__ theta, const double *__restrict__ regP,
                             const double *__restrict__ dx, const double *__restrict__ dy, double *__restrict__ r2) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) s += Ax[p] * dy[Ai[p]];
    r2[j] = (xi_d[j] + (theta[j] + regP[j]) * dx[j]) - s;
}
// multi-device mode after a refined solve: a shard stores the entries it owns (its columns of dx, its block rows of dy) into the
// lead device's job-wide vectors
__global__ void k_publish(i64 n, const char *__restrict__ col_local, const double *__restrict__ dx, double *__restrict__ dx_job,
                          i64 m, const char *__restrict__ row_local, const double *__restrict__ dy, double *__restrict__ dy_job) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && col_local[i]) dx_job[i] = dx[i];
    if (i < m && row_local[i] == 1) dy_job[i] = dy[i];
}
__global__ void k_axpy2(i64 n, double *__restrict__ x, const double *__restrict__ dxc, i64 m, double *__restrict__ y, const double *__restrict__ dyc) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += dxc[i];
    if (i < m) y[i] += dyc[i];
}

// Guarded refinement (round 5).  The unrefined normal-equations solve leaves r2 = xi_d + (theta + Rp) dx - A' dy at ROUNDING level by construction
// (dx is computed from dy) and all of its error in r1 = xi_p - A dx - Rd dy; the interior-point loops rely on that structure (a dual residual
// that a solve introduces is never removed again).  A refinement step is therefore kept only if it SHRINKS |r1|inf and leaves |r2|inf within
// 16 x of the unrefined solve's: the candidate x + c is formed beside x, its residuals are computed (they are the next step's right-hand side
// if the step is kept), the verdict is reached ON THE DEVICE (no host synchronisation inside a solve) and x is overwritten only then; after
// the first rejected step the remaining steps of the solve are no-ops.  (First version of the guard, same round: "max(|r1|, |r2|) must shrink"
// -- MPC on the north-star LP still ended with a dual residual of 0.2: steps that traded r1 for r2 passed.  gpurun_out session A.)
// ref[0] = |r1| of the current iterate, ref[1] = |r2| of the UNREFINED solve, ref[2], ref[3] = |r1|, |r2| of the candidate -- bit patterns of
// non-negative doubles (they order like the doubles; a NaN anywhere gives a pattern above +inf, i.e. "worse"); ref[4] = {rejected steps, stop},
// ref[5] = {accept, -}.
__global__ __launch_bounds__(256) void k_absmax2(i64 m, const double *__restrict__ r1, const char *__restrict__ row_mask, i64 n, const double *__restrict__ r2,
                                                 const char *__restrict__ col_mask, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long red[2][4];
    unsigned long long v1 = 0, v2 = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < m + n; i += (i64)gridDim.x * blockDim.x) {
        const bool row = i < m;
        if (row ? (row_mask && row_mask[i] != 1) : (col_mask && !col_mask[i - m])) continue;      // sharded: owned rows / columns only
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(row ? r1[i] : r2[i - m]));
        if (row) v1 = b > v1 ? b : v1; else v2 = b > v2 ? b : v2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long w1 = __shfl_xor(v1, o), w2 = __shfl_xor(v2, o);
        v1 = w1 > v1 ? w1 : v1; v2 = w2 > v2 ? w2 : v2;
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = v1; red[1][threadIdx.x >> 6] = v2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { v1 = red[0][w] > v1 ? red[0][w] : v1; v2 = red[1][w] > v2 ? red[1][w] : v2; }
        atomicMax(out, v1); atomicMax(out + 1, v2);       // maxima: the order of the blocks does not matter
    }
}
__global__ void k_refine_decide(unsigned long long *ref) {
    int *st = reinterpret_cast<int *>(ref + 4);
    const double r2_ref = __longlong_as_double((long long)ref[1]), r2_new = __longlong_as_double((long long)ref[3]);
    const bool accept = !st[1] && ref[2] < ref[0] && r2_new <= 16.0 * r2_ref;
    st[2] = accept ? 1 : 0;
    if (accept) ref[0] = ref[2];
    else if (!st[1]) { st[0] += 1; st[1] = 1; }
    ref[2] = 0; ref[3] = 0;
}
// candidate <- x + correction (kept beside x until the verdict)
__global__ void k_candidate(i64 n, const double *__restrict__ x, double *__restrict__ cx, i64 m, const double *__restrict__ y, double *__restrict__ cy) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cx[i] = x[i] + cx[i];
    if (i < m) cy[i] = y[i] + cy[i];
}
__global__ void k_refine_commit(i64 n, double *__restrict__ x, const double *__restrict__ cx, i64 m, double *__restrict__ y, const double *__restrict__ cy,
                                const unsigned long long *__restrict__ ref) {
    if (!reinterpret_cast<const int *>(ref + 4)[2]) return;
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = cx[i];
    if (i < m) y[i] = cy[i];
}

// out = own + src[0] + src[1] + ... (fixed order): the root-panel / root-rhs reduction of the multi-device mode.
// NOT in place: the ranks copy the result out of `out` at their own pace while the lead already factorises / solves
// its own copy.
__global__ void k_sum_to(i64 len, double *__restrict__ out, const double *__restrict__ own, const double *__restrict__ src, int nsrc, i64 stride) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double v = own[i];
    for (int r = 0; r < nsrc; ++r) v += src[(i64)r * stride + i];
    out[i] = v;
}

// ------------------------------------------------------------------------------------------
// Augmented system (K2): K = [-(Theta^-1 + Rp) A'; A Rd], nodes 0..n-1 = variables, n..n+m-1 = constraints
// (/root/reference/src/KKT/Cholmod/sqd.jl:24-74)
// ------------------------------------------------------------------------------------------
// D2 = [theta_inv + regP ; 1]: the vector the assembly lists refer to (diagonal of a variable node: -1 * D2[j];
// an off-diagonal entry A[i,j] * D2[n])                                                            sqd.jl:44-50
__global__ void k_k2_diag(i64 n, const double *__restrict__ theta, const double *__restrict__ regP, double *__restrict__ D2) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) D2[j] = theta[j] + regP[j];
    else if (j == n) D2[j] = 1.0;
}
// permuted right-hand side [xi_d ; xi_p]                                                          sqd.jl:62-66
// Sharded runs (node_local: 0 = another rank's node, 1 = local, 2 = node of the replicated root front): a rank loads the right-hand
// side of its own nodes, rank 0 that of the root nodes; the all-reduce of the root rhs completes them with the ranks' contributions.
__global__ void k_k2_rhs(i64 N, i64 n, const i32 *__restrict__ perm, const char *__restrict__ node_local, int rank,
                         const double *__restrict__ xi_p, const double *__restrict__ xi_d, double *__restrict__ xw) {
    const i64 kk = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (kk >= N) return;
    const i32 v = perm[kk];
    const char nl = node_local[v];
    xw[kk] = (nl == 0 || (nl == 2 && rank != 0)) ? 0.0 : ((v < n) ? xi_d[v] : xi_p[v - n]);
}
// between the sweeps of L S L' x = b:  z = S y
__global__ void k_apply_signs(i64 N, const double *__restrict__ csign, double *__restrict__ xw) {
    const i64 kk = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (kk < N) xw[kk] *= csign[kk];
}
// [dx ; dy] = P' x                                                                                 sqd.jl:69-72
// owned_only (single-process multi-device mode: dx / dy are the job-wide vectors on the lead device): a rank stores the nodes it owns
// (its own; the root nodes on rank 0) and leaves the others alone; otherwise the entries of other ranks' nodes are written as 0.
__global__ void k_k2_out(i64 N, i64 n, const i32 *__restrict__ perm, const char *__restrict__ node_local, int rank, int owned_only,
                         const double *__restrict__ xw, double *__restrict__ dx, double *__restrict__ dy) {
    const i64 kk = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (kk >= N) return;
    const i32 v = perm[kk];
    const char nl = node_local[v];
    if (owned_only && !(nl == 1 || (nl == 2 && rank == 0))) return;
    const double val = nl ? xw[kk] : 0.0;
    if (v < n) dx[v] = val; else dy[v - n] = val;
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
static inline unsigned nblk(i64 n, int b) { return (unsigned)((n + b - 1) / b); }

void launch_compute_d(hipStream_t st, i64 n, const double *theta, const double *regP, double *D) {
    if (n > 0) hipLaunchKernelGGL(k_compute_d, dim3(nblk(n, 256)), dim3(256), 0, st, n, theta, regP, D);
}
void launch_zero_panels(hipStream_t st, const DevArrays &a, int part) {
    const i64 first = (part == 1) ? a.n_zero_lower : 0, last = (part == 0) ? a.n_zero_lower : a.n_zero_tasks;
    if (last > first) hipLaunchKernelGGL(k_zero_panels, dim3((unsigned)(last - first), part == 1 ? 8u : 1u), dim3(256), 0, st, a.zero_tasks + 2 * first, a.ctx);
    if (part != 1 && a.n_zero_small > 0)     // the small panels are never `upper`
        hipLaunchKernelGGL(k_zero_small, dim3((unsigned)((a.n_zero_small + 3) / 4)), dim3(256), 0, st, a.zero_small, a.n_zero_small, a.ctx);
}
void launch_assemble(hipStream_t st, const DevArrays &a, const double *D, const double *regD, int part) {
    if (a.n_asm > 0)
        hipLaunchKernelGGL(k_assemble, dim3(nblk(a.n_asm, 256)), dim3(256), 0, st, a.n_asm, a.asm_target_small, a.asm_diag,
                           a.asm_ptr, a.pair_w, a.pair_j, D, regD, a.ctx.Lval, part < 0 ? nullptr : a.asm_upper, part);
}
void launch_single_factor(hipStream_t st, const DevArrays &a) {
    if (a.n_single > 0)
        hipLaunchKernelGGL(k_single_factor, dim3(nblk(a.n_single, 256)), dim3(256), 0, st, a.n_single, a.single_loff, a.single_dinvoff,
                           a.single_col, a.ctx.Lval, a.ctx.dinv, a.ctx.info, a.ctx.csign);
}
// rhs: 0 / 1 = that right-hand side, 2 = both in one launch (grid y)
void launch_single_solve(hipStream_t st, const DevArrays &a, int rhs) {
    if (a.n_single > 0)
        hipLaunchKernelGGL(k_single_solve, dim3(nblk(a.n_single, 256), rhs == 2 ? 2u : 1u), dim3(256), 0, st, a.n_single, a.single_dinvoff, a.single_col,
                           a.ctx.dinv, a.ctx.xw + (rhs == 1 ? a.ctx.xw2 : 0), a.ctx.xw2);
}
// nrhs = 2 (solve schedules only): the two persistent sweep kernels run their two-right-hand-side instances (one pass over L for
// both), every other solve kernel is launched once per right-hand side (the second on a context whose xw / uc point at the copies)
void launch_tasks(hipStream_t st, const DevArrays &a, const Launch &L, const SweepArgs *sw, int nrhs) {
    if (L.count <= 0) return;
    const dim3 g((unsigned)L.count);
    const bool sgn = a.ctx.csign != nullptr;                  // K2: signed Cholesky
    if (nrhs == 2) {
        if (L.kind == LK_FWD_SWEEP) {
            if (sw && L.pad) hipLaunchKernelGGL((k_fwd_sweep<2, true>), g, dim3(256), 0, st, a.fwd_sweep_tasks + L.first, a.ctx, *sw);
            else if (sw) hipLaunchKernelGGL(k_fwd_sweep<2>, g, dim3(256), 0, st, a.fwd_sweep_tasks + L.first, a.ctx, *sw);
            return;
        }
        if (L.kind == LK_BWD_SWEEP) {
            if (sw && L.pad) hipLaunchKernelGGL((k_bwd_sweep<2, true>), g, dim3(256), 0, st, a.bwd_sweep_tasks + L.first, a.ctx, *sw);
            else if (sw) hipLaunchKernelGGL(k_bwd_sweep<2>, g, dim3(256), 0, st, a.bwd_sweep_tasks + L.first, a.ctx, *sw);
            return;
        }
        // round 6: the kernels of a pair that are not sweeps run BOTH right-hand sides in one launch (grid y = right-hand side; round 5 measured that launching
        // them once per right-hand side was the whole 1.13 - 1.21 x of a pair over a single solve, profiles/r05_solve_one_group.txt).  Same arithmetic per
        // right-hand side: the pair stays bit-identical to two solves.
        const dim3 g2((unsigned)L.count, 2u);
        switch (L.kind) {
        case LK_FWD_GATHER: hipLaunchKernelGGL(k_fwd_gather, g2, dim3(256), 0, st, a.fwd_gather_tasks + L.first, a.ctx); return;
        case LK_FWD_SMALL: hipLaunchKernelGGL(k_fwd_small, g2, dim3(256), 0, st, a.fwd_small_tasks + L.first, a.ctx); return;
        case LK_BWD_SMALL: hipLaunchKernelGGL(k_bwd_small, g2, dim3(256), 0, st, a.bwd_small_tasks + L.first, a.ctx); return;
        default: break;
        }
        launch_tasks(st, a, L, sw, 1);
        DevArrays b = a;
        b.ctx.xw += a.ctx.xw2; b.ctx.uc += a.ctx.uc2;
        launch_tasks(st, b, L, sw, 1);
        return;
    }
#define TLPK_LAUNCH_S(KERNEL, TASKS) do { if (sgn) hipLaunchKernelGGL(KERNEL<true>, g, dim3(256), 0, st, TASKS + L.first, a.ctx); \
                                          else hipLaunchKernelGGL(KERNEL<false>, g, dim3(256), 0, st, TASKS + L.first, a.ctx); } while (0)
    switch (L.kind) {
    case LK_EXTEND_ADD: {
        // TLPK_EA_LDS (tuning knob): extra dynamic LDS per workgroup = fewer resident workgroups = a smaller working set of parent
        // columns (the kernel's fabric traffic is parent lines evicted between two children's contributions)
        static const unsigned ea_lds = [] { const char *e = std::getenv("TLPK_EA_LDS"); return e ? (unsigned)std::atoi(e) : 0u; }();
        hipLaunchKernelGGL(k_extend_add, g, dim3(256), ea_lds, st, a.ea_tasks + L.first, a.ctx); break;
    }
    case LK_FRONT_ASSEMBLE:
        hipLaunchKernelGGL(k_front_assemble, g, dim3(256), 0, st, a.fa_tasks + L.first, a.ctx, a.asm_colptr, a.asm_target, a.asm_diag, a.asm_ptr,
                           a.pair_w, a.pair_j, a.asm_D, a.asm_regD);
        break;
    case LK_POTRF: case LK_POTRF_WIDE: {
        // 64 x 64 diagonal-block kernel (TLPK_POTRF_MODE): 3 = potrf_block_dpp (DPP broadcasts, one wave per panel chain; the default since round 5:
        // 16.7 us per block against 32.9), 2 = potrf_block_pair (one barrier per two columns; bit-identical to 0), 0 = potrf_block (one barrier per
        // column, rounds 1..4), 1 = potrf_block_wave (readlane broadcasts; measured slower, see there).  TLPK_POTRF_WAVE / TLPK_POTRF_PAIR: the older switches.
        auto mode = [] {
            const char *m = std::getenv("TLPK_POTRF_MODE"); if (m) return std::atoi(m) & 3;
            const char *w = std::getenv("TLPK_POTRF_WAVE"); if (w && std::atoi(w) != 0) return 1;
            const char *e = std::getenv("TLPK_POTRF_PAIR"); if (e) return std::atoi(e) == 0 ? 0 : 2;
            return 3;
        };
        const bool dyn = std::getenv("TLPK_POTRF_DYN") != nullptr;             // (diagnostics / the kernel-agreement test: the mode is re-read at every launch)
        static const int pm0 = mode();
        const int pm = dyn ? mode() : pm0;
#define TLPK_LAUNCH_PM(KERNEL, SG) do { if (pm == 3) hipLaunchKernelGGL((KERNEL<SG, 3>), g, dim3(256), 0, st, a.potrf_tasks + L.first, a.ctx); \
                                        else if (pm == 1) hipLaunchKernelGGL((KERNEL<SG, 1>), g, dim3(256), 0, st, a.potrf_tasks + L.first, a.ctx); \
                                        else if (pm == 2) hipLaunchKernelGGL((KERNEL<SG, 2>), g, dim3(256), 0, st, a.potrf_tasks + L.first, a.ctx); \
                                        else hipLaunchKernelGGL((KERNEL<SG, 0>), g, dim3(256), 0, st, a.potrf_tasks + L.first, a.ctx); } while (0)
#define TLPK_LAUNCH_P(KERNEL) do { if (sgn) TLPK_LAUNCH_PM(KERNEL, true); else TLPK_LAUNCH_PM(KERNEL, false); } while (0)
        if (L.kind == LK_POTRF) TLPK_LAUNCH_P(k_potrf); else TLPK_LAUNCH_P(k_potrf_wide);
#undef TLPK_LAUNCH_PM
#undef TLPK_LAUNCH_P
        break;
    }
    case LK_POTRF_SMALL: TLPK_LAUNCH_S(k_potrf_small, a.potrf_tasks); break;
    case LK_TRSM: TLPK_LAUNCH_S(k_trsm, a.trsm_tasks); break;
    case LK_TRSM_THIN: TLPK_LAUNCH_S(k_trsm_thin, a.trsm_tasks); break;
    case LK_UPDATE: {
        // 512-thread workgroups, 2 x 4 waves of 64 x 32 sub-tiles, 4 waves per SIMD (round 3: k_update 32.3 -> 31.7 ms on C4, 101.2 -> 100.2 ms
        // on the headline instance against 2 x 2 waves of 64 x 64 at 2 waves per SIMD; profiles/r03_update_8waves.txt).  TLPK_UPD_WAVES=4: the
        // round-2 kernel.  That doubling the resident waves buys 2 % says the idle 22 % of the matrix pipes is not a latency-hiding problem.
        static const int nw = [] { const char *e = std::getenv("TLPK_UPD_WAVES"); return (e && std::atoi(e) == 4) ? 4 : 8; }();
        if (nw == 8) {
            if (sgn) hipLaunchKernelGGL((k_update<true, 8>), g, dim3(512), 0, st, a.update_tasks + L.first, a.ctx);
            else hipLaunchKernelGGL((k_update<false, 8>), g, dim3(512), 0, st, a.update_tasks + L.first, a.ctx);
        } else TLPK_LAUNCH_S(k_update, a.update_tasks);
        break;
    }
    case LK_CHAIN: {
        // Persistent: min(items, grid) workgroups loop over the launch's ticket counter.  ONE workgroup per CU by default (TLPK_CHAIN_DYNLDS = 70 000 bytes of
        // dynamic LDS that nobody uses, on top of the 81 920 static ones: a second workgroup does not fit): every role has its CU to itself -- the diagonal
        // block's serial chain runs 20 % slower beside another workgroup's matrix-core waves, an update tile nearly twice as fast alone -- measured on the
        // pds-class LP: 15.08 ms per Newton step against 16.54 with two per CU and 15.76 with the launches (profiles/r06_chain_*.txt).
        // TLPK_CHAIN_DYNLDS=0 TLPK_CHAIN_GRID=512: two per CU.  More workgroups than fit are harmless: one that starts late draws the next ticket.
        const unsigned dyn = [] { const char *e = std::getenv("TLPK_CHAIN_DYNLDS"); return e ? (unsigned)std::max(0, std::atoi(e)) : 70000u; }();
        const unsigned grid_max = [&] { const char *e = std::getenv("TLPK_CHAIN_GRID"); return e ? (unsigned)std::max(1, std::atoi(e)) : (dyn >= 70000u ? 256u : 512u); }();
        static const unsigned spin_max = [] { const char *e = std::getenv("TLPK_CHAIN_TIMEOUT_MS"); return (unsigned)(e ? std::min(600000, std::max(1, std::atoi(e))) : 2000); }();
        const ChainArgs ca{a.chain_items + L.first, (i32)L.count, a.chain_cnt, L.pad, a.update_tasks, a.reduce_tasks, a.potrf_tasks, a.trsm_tasks,
                           a.chain_trace ? a.chain_trace + 4 * L.first : nullptr, spin_max, dyn};
        const dim3 gc((unsigned)std::min<i64>(L.count, grid_max));
        // ONE dependency-driven launch at a time per device, process-wide (TLPK_CHAIN_SERIAL=0 lifts it).  Each launch is deadlock-free by itself, but several of them
        // side by side on ONE device -- eight shards of a multi-device handle on one GPU (the test configuration), independent handles of one process -- froze: with
        // 8 or 16 hardware queues the eight-shards test ended TLPK_INTERNAL in 15 - 40 % of the runs (the time stamps show ~14 workgroups of one launch, all long
        // update tiles, standing still for 0.68 s inside their role while every other workgroup of the device spins, and finishing the moment the others give up;
        // never with GPU_MAX_HW_QUEUES <= 4; profiles/r06_chain_poll_storm.txt).  The launch therefore waits for the previous chain launch enqueued on this device
        // (an event of its stream: dependencies only point back in enqueue order, so the streams' own event graph stays acyclic); launches of other kinds run
        // beside it as before.  Not inside a stream capture (hipGraph replay of single-group schedules: one handle, one launch at a time by construction).
        static const bool chain_serial = [] { const char *e = std::getenv("TLPK_CHAIN_SERIAL"); return !e || std::atoi(e) != 0; }();
        static std::mutex chain_mu;
        static hipEvent_t chain_ev[64] = {};
        static bool chain_rec[64] = {};
        int dev = 0; (void)hipGetDevice(&dev);
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone; (void)hipStreamIsCapturing(st, &cs);
        const bool ser = chain_serial && cs == hipStreamCaptureStatusNone && dev >= 0 && dev < 64;
        std::unique_lock<std::mutex> lk(chain_mu, std::defer_lock);
        if (ser) {
            lk.lock();
            if (!chain_ev[dev] && hipEventCreateWithFlags(&chain_ev[dev], hipEventDisableTiming) != hipSuccess) chain_ev[dev] = nullptr;
            if (chain_ev[dev] && chain_rec[dev]) (void)hipStreamWaitEvent(st, chain_ev[dev], 0);
        }
        if (sgn) hipLaunchKernelGGL(k_chain<true>, gc, dim3(256), dyn, st, ca, a.ctx);
        else hipLaunchKernelGGL(k_chain<false>, gc, dim3(256), dyn, st, ca, a.ctx);
        if (ser && chain_ev[dev]) { (void)hipEventRecord(chain_ev[dev], st); chain_rec[dev] = true; }
        break;
    }
    case LK_UPDATE_T64: TLPK_LAUNCH_S(k_update64, a.update_tasks); break;
    case LK_UPDATE_REDUCE: hipLaunchKernelGGL(k_update_reduce, dim3((unsigned)L.count * RED_SPLIT), dim3(256), 0, st, a.reduce_tasks + L.first, a.ctx); break;
    case LK_FWD_GATHER: hipLaunchKernelGGL(k_fwd_gather, g, dim3(256), 0, st, a.fwd_gather_tasks + L.first, a.ctx); break;
    case LK_FWD_DIAG: hipLaunchKernelGGL(k_fwd_diag, g, dim3(256), 0, st, a.fwd_diag_tasks + L.first, a.ctx); break;
    case LK_FWD_UPDATE: hipLaunchKernelGGL(k_fwd_update, g, dim3(256), 0, st, a.fwd_update_tasks + L.first, a.ctx); break;
    case LK_BWD_UPDATE: hipLaunchKernelGGL(k_bwd_update, g, dim3(256), 0, st, a.bwd_update_tasks + L.first, a.ctx); break;
    case LK_FWD_SMALL: hipLaunchKernelGGL(k_fwd_small, g, dim3(256), 0, st, a.fwd_small_tasks + L.first, a.ctx); break;
    case LK_BWD_SMALL: hipLaunchKernelGGL(k_bwd_small, g, dim3(256), 0, st, a.bwd_small_tasks + L.first, a.ctx); break;
    case LK_FWD_SWEEP:
        if (sw && L.pad) hipLaunchKernelGGL((k_fwd_sweep<1, true>), g, dim3(256), 0, st, a.fwd_sweep_tasks + L.first, a.ctx, *sw);
        else if (sw) hipLaunchKernelGGL(k_fwd_sweep<1>, g, dim3(256), 0, st, a.fwd_sweep_tasks + L.first, a.ctx, *sw);
        break;
    case LK_BWD_SWEEP:
        if (sw && L.pad) hipLaunchKernelGGL((k_bwd_sweep<1, true>), g, dim3(256), 0, st, a.bwd_sweep_tasks + L.first, a.ctx, *sw);
        else if (sw) hipLaunchKernelGGL(k_bwd_sweep<1>, g, dim3(256), 0, st, a.bwd_sweep_tasks + L.first, a.ctx, *sw);
        break;
    default: break;
    }
#undef TLPK_LAUNCH_S
}
void launch_k2_diag(hipStream_t st, i64 n, const double *theta, const double *regP, double *D2) {
    hipLaunchKernelGGL(k_k2_diag, dim3(nblk(n + 1, 256)), dim3(256), 0, st, n, theta, regP, D2);
}
void launch_k2_rhs(hipStream_t st, const DevArrays &a, i64 n, const double *xi_p, const double *xi_d, int rhs, int rank) {
    if (a.m > 0) hipLaunchKernelGGL(k_k2_rhs, dim3(nblk(a.m, 256)), dim3(256), 0, st, a.m, n, a.perm, a.row_local, rank, xi_p, xi_d, a.ctx.xw + (rhs ? a.ctx.xw2 : 0));
}
void launch_apply_signs(hipStream_t st, const DevArrays &a, int rhs) {
    if (a.m > 0) hipLaunchKernelGGL(k_apply_signs, dim3(nblk(a.m, 256)), dim3(256), 0, st, a.m, a.ctx.csign, a.ctx.xw + (rhs ? a.ctx.xw2 : 0));
}
void launch_k2_out(hipStream_t st, const DevArrays &a, i64 n, double *dx, double *dy, int rhs, int rank, int owned_only) {
    if (a.m > 0) hipLaunchKernelGGL(k_k2_out, dim3(nblk(a.m, 256)), dim3(256), 0, st, a.m, n, a.perm, a.row_local, rank, owned_only, a.ctx.xw + (rhs ? a.ctx.xw2 : 0), dx, dy);
}
void launch_rhs(hipStream_t st, const DevArrays &a, const double *D, const double *xi_p, const double *xi_d, int rank, int rhs) {
    if (a.m > 0)
    {
        double *w = a.rhs_w + (rhs ? a.n : 0);
        if (a.n > 0) hipLaunchKernelGGL(k_rhs_scale, dim3(nblk(a.n, 256)), dim3(256), 0, st, a.n, D, xi_d, a.col_local, w, xi_d);
        hipLaunchKernelGGL(k_rhs, dim3(nblk(a.m * 8, 256)), dim3(256), 0, st, a.m, a.perm, a.Pp, a.Pj, a.Px, w, xi_p, xi_d,
                           a.row_local, a.col_local, rank, a.ctx.xw + (rhs ? a.ctx.xw2 : 0), xi_p, (i64)0, (i64)0);
    }
}
// both right-hand sides of a pair in one launch each (grid y)
void launch_rhs2(hipStream_t st, const DevArrays &a, const double *D, const double *const *xi_p, const double *const *xi_d, int rank) {
    if (a.m > 0)
    {
        if (a.n > 0) hipLaunchKernelGGL(k_rhs_scale, dim3(nblk(a.n, 256), 2), dim3(256), 0, st, a.n, D, xi_d[0], a.col_local, a.rhs_w, xi_d[1]);
        hipLaunchKernelGGL(k_rhs, dim3(nblk(a.m * 8, 256), 2), dim3(256), 0, st, a.m, a.perm, a.Pp, a.Pj, a.Px, a.rhs_w, xi_p[0], xi_d[0],
                           a.row_local, a.col_local, rank, a.ctx.xw, xi_p[1], a.n, a.ctx.xw2);
    }
}
void launch_unpermute2(hipStream_t st, const DevArrays &a, double *const *dy, int rank) {
    if (a.m > 0) hipLaunchKernelGGL(k_unpermute, dim3(nblk(a.m, 256), 2), dim3(256), 0, st, a.m, a.perm, a.row_local, a.ctx.xw, dy[0], (double *)nullptr, rank, dy[1], a.ctx.xw2);
}
void launch_dx2(hipStream_t st, const DevArrays &a, const double *D, double *const *dy, const double *const *xi_d, double *const *dx) {
    if (a.n > 0) hipLaunchKernelGGL(k_dx, dim3(nblk(a.n, 256), 2), dim3(256), 0, st, a.n, a.Ap, a.Ai, a.Ax, D, dy[0], xi_d[0], a.col_local, dx[0], 0, dy[1], xi_d[1], dx[1]);
}
// reduce-scatter step of the multi-device reductions: this shard's slice, summed over ALL ranks in rank order (its own contribution at position
// own_rank, the peers' slices from the staging area: rank s at slot s, or s - 1 behind own_rank) -- every slice gets the same order whoever owns it
__global__ void k_sum_ranked(i64 len, double *__restrict__ inout, const double *__restrict__ stage, int nranks, int own_rank, i64 stride) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double v = 0.0;
    for (int s_ = 0; s_ < nranks; ++s_) v += (s_ == own_rank) ? inout[i] : stage[(i64)(s_ < own_rank ? s_ : s_ - 1) * stride + i];
    inout[i] = v;
}
void launch_sum_ranked(hipStream_t st, i64 len, double *inout, const double *stage, int nranks, int own_rank, i64 stride) {
    if (len > 0) hipLaunchKernelGGL(k_sum_ranked, dim3(nblk(len, 256)), dim3(256), 0, st, len, inout, stage, nranks, own_rank, stride);
}
void launch_sum_to(hipStream_t st, i64 len, double *out, const double *own, const double *src, int nsrc, i64 stride) {
    if (len > 0) hipLaunchKernelGGL(k_sum_to, dim3(nblk(len, 256)), dim3(256), 0, st, len, out, own, src, nsrc, stride);
}
void launch_unpermute(hipStream_t st, const DevArrays &a, double *dy, double *dy_shared, int rank, int rhs) {
    if (a.m > 0) hipLaunchKernelGGL(k_unpermute, dim3(nblk(a.m, 256)), dim3(256), 0, st, a.m, a.perm, a.row_local, a.ctx.xw + (rhs ? a.ctx.xw2 : 0), dy, dy_shared, rank, dy, (i64)0);
}
void launch_residuals(hipStream_t st, const DevArrays &a, const double *xi_p, const double *xi_d, const double *theta, const double *regP,
                      const double *regD, const double *dx, const double *dy, double *r1, double *r2, int rank, int xip_all) {
    if (a.m > 0) hipLaunchKernelGGL(k_resid_rows, dim3(nblk(a.m, 256)), dim3(256), 0, st, a.m, a.Tp, a.Tj, a.Tx, xi_p, regD, dx, dy, a.row_local, rank, xip_all, r1);
    if (a.n > 0) hipLaunchKernelGGL(k_resid_cols, dim3(nblk(a.n, 256)), dim3(256), 0, st, a.n, a.Ap, a.Ai, a.Ax, xi_d, theta, regP, dx, dy, r2);
}
void launch_publish(hipStream_t st, const DevArrays &a, const double *dx, double *dx_job, const double *dy, double *dy_job) {
    const i64 len = std::max(a.n, a.m);
    if (len > 0) hipLaunchKernelGGL(k_publish, dim3(nblk(len, 256)), dim3(256), 0, st, a.n, a.col_local, dx, dx_job, a.m, a.row_local, dy, dy_job);
}
void launch_axpy2(hipStream_t st, i64 n, double *x, const double *dxc, i64 m, double *y, const double *dyc) {
    const i64 len = std::max(n, m);
    if (len > 0) hipLaunchKernelGGL(k_axpy2, dim3(nblk(len, 256)), dim3(256), 0, st, n, x, dxc, m, y, dyc);
}
void launch_absmax2(hipStream_t st, const DevArrays &a, const double *r1, const double *r2, unsigned long long *out, int owned_only) {
    const i64 len = a.m + a.n;
    if (len > 0) hipLaunchKernelGGL(k_absmax2, dim3((unsigned)std::min<i64>(nblk(len, 256), 1024)), dim3(256), 0, st, a.m, r1, owned_only ? a.row_local : nullptr,
                                    a.n, r2, owned_only ? a.col_local : nullptr, out);
}
void launch_refine_decide(hipStream_t st, unsigned long long *ref) { hipLaunchKernelGGL(k_refine_decide, dim3(1), dim3(1), 0, st, ref); }
void launch_candidate(hipStream_t st, i64 n, const double *x, double *cx, i64 m, const double *y, double *cy) {
    const i64 len = std::max(n, m);
    if (len > 0) hipLaunchKernelGGL(k_candidate, dim3(nblk(len, 256)), dim3(256), 0, st, n, x, cx, m, y, cy);
}
void launch_refine_commit(hipStream_t st, i64 n, double *x, const double *cx, i64 m, double *y, const double *cy, const unsigned long long *ref) {
    const i64 len = std::max(n, m);
    if (len > 0) hipLaunchKernelGGL(k_refine_commit, dim3(nblk(len, 256)), dim3(256), 0, st, n, x, cx, m, y, cy, ref);
}
void launch_dx(hipStream_t st, const DevArrays &a, const double *D, const double *dy, const double *xi_d, double *dx, int local_only) {
    if (a.n > 0) hipLaunchKernelGGL(k_dx, dim3(nblk(a.n, 256)), dim3(256), 0, st, a.n, a.Ap, a.Ai, a.Ax, D, dy, xi_d, a.col_local, dx, local_only, dy, xi_d, dx);
}

}  // namespace tlpk
