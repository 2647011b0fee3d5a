// tlpk_device.hpp -- device-side views shared by kernels.hip and tlpk_api.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include "tlpk_host.hpp"

namespace tlpk {

// passed by value to every front kernel
struct DevCtx {
    const FrontDesc *fronts;
    const i32 *rowidx;
    const i32 *rel;
    const i32 *ea_tab;  // extend-add lookup (FrontDesc.eatab)
    const i32 *children;
    const i64 *gth_ptr, *gth_src;   // forward gather lists (per front row: the children's uc entries)
    double *Lval;       // supernodal panels of L
    double *U0, *U1;    // ping-pong update-matrix buffers (by tree depth parity)
    double *uc;         // solve contribution vectors
    double *xw;         // permuted right-hand side / solution
    double *dinv;       // inverses of the NB_IN x NB_IN diagonal blocks of L (written by k_potrf)
    double *spart;      // split-K scratch: one TILE x TILE partial product per slot
    int *info;          // info[0] = smallest failing pivot column (INT_MAX = none); info[1] != 0: a sweep gave up waiting
    int upd_remap;         // k_update blockIdx -> task mapping (0 identity, 1 XCD-contiguous, 2 runs of 64 tasks per XCD)
    const double *csign;   // K2 (augmented system): +1 / -1 per permuted column, the S of P K P' = L S L'; nullptr for K1
    const i32 *upd_seg;    // K-segment lists of the update tasks (UpdateTask.seg)
    int small_full;        // TLPK_SMALL_FULL=1 (debugging): the small-front solve kernels always take their (16 columns, 256 rows) body
    i64 xw2, uc2;          // two-right-hand-side solves: the second rhs / solution at xw + xw2, its contribution vectors at uc + uc2
};

// per-launch arguments of the persistent sweep kernels
struct SweepArgs {
    unsigned long long *ticket;   // hand-out counter of this launch: reset to all ones with the hand-over words (ONE memset per solve covers
                                  // both), so that atomicAdd(ticket, 1) + 1 hands out 0, 1, 2, ... -- no per-launch host state in the kernel
                                  // arguments: the solve schedule can be replayed from a captured graph
    double *xh;                   // hand-over words of this direction, one per permuted column, sentinel-filled before the solve
    i64 xh2;                      // ... of the second right-hand side at xh + xh2 (two-rhs sweeps)
    int poll_fast, poll_nfast, poll_slow;   // polling back-off (units of s_sleep 1 = 64 clocks): first poll_nfast polls every poll_fast, then every poll_slow
    const SolveTask *small = nullptr;       // round 6: the small-front tasks of this direction (items with slot == 2 of a merged launch name groups of four of them)
};

struct DevArrays {
    DevCtx ctx{};
    i64 m = 0, n = 0;
    // A (CSC + CSR)
    i64 *Ap = nullptr; i32 *Ai = nullptr; double *Ax = nullptr;
    i64 *Tp = nullptr; i32 *Tj = nullptr; double *Tx = nullptr;
    i32 *perm = nullptr;
    i64 *Pp = nullptr; i32 *Pj = nullptr; double *Px = nullptr;   // K1: CSR of A with the rows in permuted order (k_rhs)
    double *rhs_w = nullptr;                  // D .* xi_d of the current solve (k_rhs_scale)
    i32 *zero_tasks = nullptr; i64 n_zero_tasks = 0;   // (front, c0) pairs of k_zero_panels
    i64 n_zero_lower = 0;                              // the first n_zero_lower pairs: fronts that are not `upper` (symbolic.cpp step 13d)
    unsigned char *asm_upper = nullptr; bool has_upper = false;   // per assembled entry: 1 = its front is upper
    i32 *zero_small = nullptr; i64 n_zero_small = 0;   // fronts zeroed whole, one wave each
    char *row_local = nullptr, *col_local = nullptr;
    // assembly lists (local entries only)
    i64 n_asm = 0;
    i64 *asm_target = nullptr; i32 *asm_diag = nullptr; i64 *asm_ptr = nullptr;
    double *pair_w = nullptr; i32 *pair_j = nullptr;
    // task arrays
    FaTask *fa_tasks = nullptr;              // k_front_assemble tiles
    i64 *asm_colptr = nullptr;               // per permuted column: first entry of the (compacted) assembly list; k_front_assemble walks the columns of its tile
    i64 *asm_target_small = nullptr;         // asm_target with -1 for the entries k_front_assemble forms: what k_assemble writes
    const double *asm_D = nullptr, *asm_regD = nullptr;   // handle-owned D = 1 / (theta + regP) (K2: D2) and regD, read by the assembly kernels
    EaTask *ea_tasks = nullptr; PotrfTask *potrf_tasks = nullptr; TrsmTask *trsm_tasks = nullptr;
    UpdateTask *update_tasks = nullptr, *reduce_tasks = nullptr;
    ChainItem *chain_items = nullptr;                 // items of the LK_CHAIN launches (k_chain)
    unsigned long long *chain_trace = nullptr;        // TLPK_CHAIN_TRACE=1: 4 time stamps per item (kernels.hip: k_chain), read back through tlpk_symbolic_get("chain_trace")
    unsigned *chain_cnt = nullptr; i64 n_chain_cnt = 0;   // their tickets + completion counters, zeroed at the start of every update
    i64 n_single = 0; i64 *single_loff = nullptr, *single_dinvoff = nullptr; i32 *single_col = nullptr;   // isolated 1 x 1 fronts
    SolveTask *fwd_gather_tasks = nullptr, *fwd_diag_tasks = nullptr, *fwd_update_tasks = nullptr,
              *bwd_update_tasks = nullptr, *fwd_small_tasks = nullptr, *bwd_small_tasks = nullptr,
              *fwd_sweep_tasks = nullptr, *bwd_sweep_tasks = nullptr;
    unsigned long long *sweep_tickets = nullptr;      // one counter per sweep launch of the schedules, stored right in front of ...
    double *sweep_xh = nullptr;                       // ... the hand-over words: [0, m) forward sweep, [m, 2m) backward sweep
    i64 sweep_reset_bytes = 0, sweep_reset_bytes2 = 0;   // tickets + hand-over words (of one / two right-hand sides): one hipMemsetAsync(0xFF) per solve
};

void launch_compute_d(hipStream_t st, i64 n, const double *theta, const double *regP, double *D);
// part: -1 = everything, 0 = the lower fronts only, 1 = the upper fronts only (symbolic.cpp step 13d)
void launch_assemble(hipStream_t st, const DevArrays &a, const double *D, const double *regD, int part = -1);
void launch_zero_panels(hipStream_t st, const DevArrays &a, int part = -1);
void launch_tasks(hipStream_t st, const DevArrays &a, const Launch &L, const SweepArgs *sw = nullptr, int nrhs = 1);
void launch_single_factor(hipStream_t st, const DevArrays &a);
void launch_single_solve(hipStream_t st, const DevArrays &a, int rhs = 0);
void launch_rhs(hipStream_t st, const DevArrays &a, const double *D, const double *xi_p, const double *xi_d, int rank, int rhs = 0);
void launch_unpermute(hipStream_t st, const DevArrays &a, double *dy, double *dy_shared = nullptr, int rank = 0, int rhs = 0);
// the pair's per-right-hand-side kernels, both right-hand sides in ONE launch each (grid y = right-hand side): single-rank handles
void launch_rhs2(hipStream_t st, const DevArrays &a, const double *D, const double *const *xi_p, const double *const *xi_d, int rank);
void launch_unpermute2(hipStream_t st, const DevArrays &a, double *const *dy, int rank);
void launch_dx2(hipStream_t st, const DevArrays &a, const double *D, double *const *dy, const double *const *xi_d, double *const *dx);
void launch_residuals(hipStream_t st, const DevArrays &a, const double *xi_p, const double *xi_d, const double *theta, const double *regP,
                      const double *regD, const double *dx, const double *dy, double *r1, double *r2, int rank, int xip_all = 0);
void launch_publish(hipStream_t st, const DevArrays &a, const double *dx, double *dx_job, const double *dy, double *dy_job);
void launch_axpy2(hipStream_t st, i64 n, double *x, const double *dxc, i64 m, double *y, const double *dyc);
// guarded refinement (kernels.hip: k_absmax2 ...): max-norm of (r1, r2) into *out (bit pattern, atomicMax: zero it first), verdict, candidate, commit
void launch_absmax2(hipStream_t st, const DevArrays &a, const double *r1, const double *r2, unsigned long long *out, int owned_only = 0);
void launch_refine_decide(hipStream_t st, unsigned long long *ref);
void launch_candidate(hipStream_t st, i64 n, const double *x, double *cx, i64 m, const double *y, double *cy);
void launch_refine_commit(hipStream_t st, i64 n, double *x, const double *cx, i64 m, double *y, const double *cy, const unsigned long long *ref);
void launch_dx(hipStream_t st, const DevArrays &a, const double *D, const double *dy, const double *xi_d, double *dx, int local_only = 0);
void launch_sum_to(hipStream_t st, i64 len, double *out, const double *own, const double *src, int nsrc, i64 stride);
void launch_sum_ranked(hipStream_t st, i64 len, double *inout, const double *stage, int nranks, int own_rank, i64 stride);
void launch_k2_diag(hipStream_t st, i64 n, const double *theta, const double *regP, double *D2);
void launch_k2_rhs(hipStream_t st, const DevArrays &a, i64 n, const double *xi_p, const double *xi_d, int rhs = 0, int rank = 0);
void launch_apply_signs(hipStream_t st, const DevArrays &a, int rhs = 0);
void launch_k2_out(hipStream_t st, const DevArrays &a, i64 n, double *dx, double *dy, int rhs = 0, int rank = 0, int owned_only = 0);

}  // namespace tlpk
