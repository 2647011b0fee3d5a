/*
 * tlpk.h -- C ABI of libtlpk.so: the MI355X (gfx950) backend for Tulip's normal-equations
 * Newton step.  This is the drop-in boundary: a Julia `HIPNormalEquations <: AbstractKKTSolver`
 * (tulip.jl_amd/julia/hip.jl) binds these entry points with `ccall`.
 *
 * Reference interface replaced (citations into /root/reference):
 *   tlpk_create   <-> KKT.setup(A, ::K1, backend)            src/KKT/KKT.jl:59, Cholmod/spd.jl:5-20
 *   tlpk_update   <-> KKT.update!(kkt, θinv, regP, regD)      src/KKT/KKT.jl:65-83, Cholmod/spd.jl:22-50
 *   tlpk_solve    <-> KKT.solve!(dx, dy, kkt, ξp, ξd)         src/KKT/KKT.jl:85-100, Cholmod/spd.jl:52-70
 *   tlpk_backend_name / tlpk_system_name <-> KKT.backend / KKT.linear_system   KKT.jl:107-121
 *   tlpk_destroy  <-> GC finalizer of the solver object
 *
 * Conventions: plain pointers and sizes only; every function returns a TLPK_* code and never
 * throws, aborts or retains a host pointer after it returns.  Host-pointer entry points block
 * until results are in host memory.  One handle is used by one thread at a time; several
 * handles may coexist.  Numeric factorise + solves run on the device; the analyse phase
 * (ordering, elimination tree, supernodes, schedules) runs on the host inside tlpk_create.
 */
#ifndef TLPK_H
#define TLPK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tlpk_handle tlpk_handle;

/* return codes */
#define TLPK_OK 0
#define TLPK_NOT_POSDEF 1   /* -> PosDefException in the glue (spd.jl:47); handle stays usable */
#define TLPK_BADARG 2       /* -> DimensionMismatch / ArgumentError (spd.jl:26-34) */
#define TLPK_OOM 3          /* -> OutOfMemoryError (HSD.jl:327-329) */
#define TLPK_HIPERR 4       /* HIP runtime error; tlpk_last_error() has the text */
#define TLPK_NO_DEVICE 5    /* numeric call on an analyse-only handle, or no GPU visible */
#define TLPK_TOO_LARGE 6    /* symbolic nnz(L) exceeds the memory budget (SURVEY.md App. C gate) */
#define TLPK_NOT_FACTORED 7 /* solve before a successful update */
#define TLPK_INTERNAL 8

/* linear system (KKT.jl:37-56 / systems.jl) */
#define TLPK_SYSTEM_K1 0     /* normal equations  A D A' + Rd, Cholesky            (Cholmod/spd.jl) */
#define TLPK_SYSTEM_K2 1     /* augmented system [-(Theta^-1+Rp) A'; A Rd], signed Cholesky L S L' = the LDL' of a
                                quasi-definite matrix without pivoting (Cholmod/sqd.jl, LDLFactorizations/ldlfact.jl);
                                row_block gives per-block ordering, stream groups and a root front; sharded / multi-device handles
                                replicate the root front (pivots of both signs), its assembled entries come from rank 0 */

/* ordering selector */
#define TLPK_ORDER_AMD 0
#define TLPK_ORDER_NATURAL 1
#define TLPK_ORDER_USER 2

typedef struct tlpk_options {
    int32_t struct_size;       /* = sizeof(tlpk_options); set by tlpk_default_options */
    int32_t device;            /* HIP device ordinal; -1 = analyse only (no device is touched) */
    int32_t ordering;          /* TLPK_ORDER_* */
    int32_t relax;             /* supernode amalgamation: 0 = fundamental only, 1 = relaxed */
    int32_t profile;           /* 1 = time every kernel class with HIP events (tlpk_kernel_times) */
    int32_t rank, nranks;      /* block-angular sharding over ranks (nranks = 1: everything local) */
    int32_t streams;           /* concurrent stream groups for block-angular LPs: 0 = auto (2), 1 = single group; each group also owns a side stream */
    const int64_t *user_perm;  /* TLPK_ORDER_USER: perm[new] = old, in index_base, length m */
    const int64_t *row_block;  /* block-angular hook (length m): block id >= 0, or -1 for a linking
                                  row; NULL = general sparse.  Blocks are ordered independently,
                                  linking rows last as one dense root supernode. */
    int64_t mem_budget_bytes;  /* 0 = 90 % of the device's free memory (or unlimited if device=-1) */
    int32_t system;            /* TLPK_SYSTEM_K1 (default) | TLPK_SYSTEM_K2 */
    int32_t refine_steps;      /* iterative-refinement steps per solve on the residuals of the augmented system (each step = one more
                                  pair of sweeps); 0 = none = the reference's behaviour (spd.jl:68 leaves it as a TODO).  K1; one rank or a
                                  tlpk_create_multi handle (sharded handles: tlpk_refine_local / tlpk_refine_finish, the caller owns the collective) */
    int32_t detect_blocks;     /* 1 (and row_block == NULL): find the block-angular structure of THIS matrix with tlpk_detect_blocks --
                                  the hook that survives Tulip's presolve, which renumbers the rows before KKT.setup sees them
                                  (model.jl:88-131).  No structure found: general sparse path (tlpk_create) / TLPK_BADARG (tlpk_create_multi) */
    int32_t keep_on_too_large; /* 1: tlpk_create returns a LIVE analyse-only handle together with TLPK_TOO_LARGE (tlpk_info / tlpk_last_error then describe what did not
                                  fit; the caller must destroy it).  0 (default): no handle on any failure -- the message is in tlpk_last_create_error() */
    int64_t max_link_rows;     /* detect_blocks: most linking rows to accept; 0 = max(64, m / 20) */
} tlpk_options;

typedef struct tlpk_stats {
    int64_t m, n, nnzA;
    int64_t nnzS;              /* lower triangle of A*D*A' + Rd, incl. diagonal */
    int64_t nnzL;              /* nnz of the Cholesky factor (incl. diagonal) */
    int64_t nnzL_stored;       /* doubles stored in supernodal panels (>= nnzL) */
    double  flops_chol;        /* sum_j l_j^2 (CHOLMOD `fl` convention) */
    double  flops_panel;       /* flops executed by the dense panel kernels (incl. padding zeros) */
    int64_t n_supernodes;
    int64_t n_levels;          /* depth of the supernodal elimination tree */
    int64_t max_front;         /* largest front order */
    int64_t n_pairs;           /* products in the A*D*A' assembly lists */
    int64_t device_bytes;      /* device memory held by this handle */
    int64_t launches_update, launches_solve;
    int64_t fail_col;          /* permuted column of the first non-positive pivot, or -1 */
    double  ms_analyse;        /* host analyse time */
    double  ms_last_update;    /* device time of the last update (HIP events, handle stream) */
    double  ms_last_solve;     /* device time of the last solve */
    int32_t n_local_blocks, n_blocks;
    int64_t root_panel_len;    /* doubles in the root (linking) panel reduced across ranks */
    double  flops_update;      /* flops EXECUTED by the fp64-MFMA update kernel per factorisation on the amalgamated
                                  (zero-padded) structure: 2*K*(lower-triangle target entries), summed over its launches */
    double  flops_update_alg;  /* ALGORITHMIC flops of that kernel: the share of flops_chol = sum_j l_j^2 whose target
                                  column lies outside column j's own 256-wide block column, sum_j (l_j - r_j)^2 with the
                                  true column counts l_j (no amalgamation zeros); <= flops_chol, <= flops_update */
    double  ms_enqueue_update; /* multi-device handles: host time from the entry of the last tlpk_update until the work of EVERY shard
                                  (root fronts included) was enqueued; ms_last_update is then the wall time of the whole call */
    int64_t refine_rejected;   /* refine_steps > 0: refinement steps of the last completed solve that did NOT shrink |r1|inf, or lifted |r2|inf beyond 16 x its
                                  value after the unrefined solve, and were discarded (a rejected step ends the refinement of that solve's RESULT; on one device the remaining steps are still enqueued -- solve, candidate, residuals, norms: the
                                  verdict is taken on the device without a host round trip -- and only their commit is skipped: a rejected step does not make the call cheaper); valid after tlpk_sync / a blocking solve */
    double  flops_update_chain;     /* round 6: the share of flops_update / flops_update_alg whose tiles run as items of the dependency-driven launches */
    double  flops_update_alg_chain; /* (k_chain: fronts with more than one block column on levels with few such fronts) instead of in k_update launches */
    int64_t chain_launches, chain_items;   /* number of those launches per factorisation and the items (update tiles, diagonal blocks, solve strips, reductions) they hold */
} tlpk_stats;

/* per-kernel-class timing, filled when options.profile = 1 */
#define TLPK_KC_ASSEMBLE 0
#define TLPK_KC_EXTEND_ADD 1
#define TLPK_KC_POTRF 2
#define TLPK_KC_TRSM 3
#define TLPK_KC_UPDATE 4     /* fp64-MFMA panel update (the dominant kernel) */
#define TLPK_KC_SOLVE_FWD 5
#define TLPK_KC_SOLVE_BWD 6
#define TLPK_KC_SPMV 7
#define TLPK_KC_UPDATE_REDUCE 8   /* split-K: ordered sum of partial tiles + application to the targets */
#define TLPK_KC_CHAIN 9           /* round 6: dependency-driven launches (k_chain): update tiles + diagonal blocks + triangular solves of a level's multi-block-column fronts */
#define TLPK_KC_COUNT 10
typedef struct tlpk_kernel_times {
    double  ms[TLPK_KC_COUNT];       /* summed duration of the class in the last update+solve */
    int64_t launches[TLPK_KC_COUNT];
} tlpk_kernel_times;

void tlpk_default_options(tlpk_options *opt);

/* A is m x n CSC with int64 indices (Julia SparseMatrixCSC{Float64,Int}); index_base in {0,1}.
 * A is copied; nothing is retained.  Runs the whole analyse phase and uploads the symbolic
 * structures.  Does NOT perform the throw-away numeric factorisation of spd.jl:14-17.
 * Return value != TLPK_OK: *out = NULL and tlpk_last_create_error() holds the diagnostic (for TLPK_TOO_LARGE: the bytes needed against the
 * budget, and -- K1 with a dense column of A -- the hint that KKT_System = K2 does not form A*D*A').  Only with opt->keep_on_too_large = 1 does
 * TLPK_TOO_LARGE return a live analyse-only handle (tlpk_info: symbolic nnz(L) ...), which the caller destroys. */
int tlpk_create(tlpk_handle **out, int64_t m, int64_t n, const int64_t *colptr,
                const int64_t *rowval, const double *nzval, int index_base,
                const tlpk_options *opt);
void tlpk_destroy(tlpk_handle *h);

/* Host-pointer entry points (what the Julia glue calls). */
int tlpk_update(tlpk_handle *h, const double *theta_inv /*n*/, const double *regP /*n*/,
                const double *regD /*m*/);
int tlpk_solve(tlpk_handle *h, double *dx /*n*/, double *dy /*m*/, const double *xi_p /*m*/,
               const double *xi_d /*n*/);

/* Device-pointer entry points: same semantics, arguments are device pointers on the handle's
 * device, work is enqueued on the handle's stream; tlpk_sync waits and returns the status
 * (TLPK_NOT_POSDEF is reported by tlpk_update_device itself: it reads back one status word). */
int tlpk_update_device(tlpk_handle *h, const double *d_theta_inv, const double *d_regP,
                       const double *d_regD);
int tlpk_solve_device(tlpk_handle *h, double *d_dx, double *d_dy, const double *d_xi_p,
                      const double *d_xi_d);
/* tlpk_update_device without the wait for its status word.  Everything is enqueued; on block-angular handles the root (linking) front --
 * a dense front of a few hundred columns whose factorisation is a serial chain of diagonal blocks, ~1.5 ms with the chip nearly idle --
 * goes to a stream of its own, so that the block-level forward sweeps of the next solve overlap it.  The verdict arrives with the next
 * tlpk_sync (TLPK_NOT_POSDEF; the handle stays usable); solves enqueued in between are speculative.  Unsharded handles; falls back to the
 * blocking call where there is nothing to overlap (no root front, profile mode, graph replay).  MEASURED SLOWER than the blocking call on
 * the bench workloads (C4 +1.1 ms, north-star instance +1.9 ms per step: the root front's ~20 dependent small launches queue behind the
 * chip-filling sweep workgroups; profiles/r03_async_update.txt) -- kept as an option, not used by default. */
int tlpk_update_device_async(tlpk_handle *h, const double *d_theta_inv, const double *d_regP, const double *d_regD);
/* Two right-hand sides in one pass over the factor (the solve sweeps are HBM-bound on the bytes of L: the pair costs little more than
 * one solve).  Same semantics and bit-identical results as two tlpk_solve_device calls.  Single-rank handles.  Tulip's HSD iteration
 * has such a pair: the h-system and the predictor (HSD/step.jl:63,79); tlpk_ipm_hsolve_newton uses it. */
int tlpk_solve2_device(tlpk_handle *h, double *d_dx0, double *d_dy0, const double *d_xi_p0, const double *d_xi_d0,
                       double *d_dx1, double *d_dy1, const double *d_xi_p1, const double *d_xi_d1);
int tlpk_sync(tlpk_handle *h);
void *tlpk_stream(tlpk_handle *h);           /* hipStream_t the kernels are launched on */

/* Block-angular sharding (nranks > 1): split-phase calls so that the caller owns the
 * collective (RCCL all-reduce over xGMI through whatever communicator it has).
 *   update : tlpk_update_local -> allreduce(sum) of tlpk_root_panel -> tlpk_update_finish
 *   solve  : tlpk_solve_local  -> allreduce(sum) of tlpk_root_rhs   -> tlpk_solve_finish
 * With nranks = 1 the split calls compose to exactly tlpk_update_device / tlpk_solve_device. */
int tlpk_update_local(tlpk_handle *h, const double *d_theta_inv, const double *d_regP,
                      const double *d_regD);
int tlpk_root_panel(tlpk_handle *h, double **d_ptr, int64_t *count);
int tlpk_update_finish(tlpk_handle *h);
int tlpk_solve_local(tlpk_handle *h, const double *d_xi_p, const double *d_xi_d);
int tlpk_root_rhs(tlpk_handle *h, double **d_ptr, int64_t *count);
int tlpk_solve_finish(tlpk_handle *h, double *d_dx, double *d_dy, const double *d_xi_d);
/* The pair of tlpk_solve2_device in split-phase form (two right-hand sides, one pass over this rank's part of the factor):
 *   tlpk_solve2_local -> allreduce(sum) of tlpk_root_rhs AND of tlpk_root_rhs2 -> tlpk_solve2_finish
 * (the two root right-hand sides are separate buffers of tlpk_root_rhs's length).  Bit-identical to two split solves. */
int tlpk_solve2_local(tlpk_handle *h, const double *d_xi_p0, const double *d_xi_d0, const double *d_xi_p1, const double *d_xi_d1);
int tlpk_root_rhs2(tlpk_handle *h, double **d_ptr, int64_t *count);
int tlpk_solve2_finish(tlpk_handle *h, double *d_dx0, double *d_dy0, const double *d_xi_d0, double *d_dx1, double *d_dy1, const double *d_xi_d1);
/* Iterative refinement on a sharded handle (K1), one step = one more split solve on the residuals of the augmented system
 * (the equations of /root/reference/src/KKT/KKT.jl:70-75; the reference leaves refinement as a TODO, src/KKT/Cholmod/spd.jl:68):
 *   tlpk_refine_local(h, dx, dy, xi_p, xi_d) -> allreduce(sum) of tlpk_root_rhs -> tlpk_refine_finish(h, dx, dy)
 * after a finished solve, as often as wanted; dx / dy in the layout tlpk_solve_finish leaves (a rank's own columns and block rows,
 * the linking rows replicated) are corrected in place.  Every rank forms the residuals of the rows / columns it owns and its PARTIAL
 * sums of the linking rows; the reduction inside the solve completes them.  tlpk_options.refine_steps does this inside
 * tlpk_solve_device (one rank) and inside tlpk_solve of a tlpk_create_multi handle (the library owns the reductions there); on a
 * sharded handle the caller owns the collective, hence the split form.  With nranks = 1 the two calls compose to exactly one
 * refinement step of tlpk_solve_device. */
int tlpk_refine_local(tlpk_handle *h, const double *d_dx, const double *d_dy, const double *d_xi_p, const double *d_xi_d);
int tlpk_refine_finish(tlpk_handle *h, double *d_dx, double *d_dy);
/* Copy the root panel (which = 0), the root rhs (which = 1) or the second root rhs of a pair (which = 2) out of (dir = 0) / into (dir = 1) a
 * caller-owned device buffer, on the handle's stream -- for callers whose communicator wants to
 * own the memory it reduces (torch.distributed tensors). */
int tlpk_root_copy(tlpk_handle *h, int which, int dir, double *d_buf);

/* Single-process multi-GPU (block-angular LPs only): ONE handle, driven by one host thread, shards the diagonal
 * blocks over `ngpus` devices of this node -- what a Julia process needs (`TlpHIP.Backend(row_block = :auto, ngpus = 8)`).
 * Needs `opt->row_block` or `opt->detect_blocks`; K1 or K2; `opt->refine_steps` is honoured (K1).  devices: ngpus HIP ordinals, or
 * NULL for 0 .. ngpus-1 (an ordinal may repeat: several shards on one device, for testing).  One host analyse serves all shards; the
 * two reductions of a Newton step (root panel, root right-hand side) are done inside the library, stream-ordered -- peer-to-peer
 * copies + an ordered sum on devices[0], or ncclAllReduce (librccl.so through dlopen) with TLPK_MULTI_REDUCE=rccl --; results are
 * gathered on devices[0].  The handle accepts tlpk_update / tlpk_solve / tlpk_sync / tlpk_info / tlpk_get_perm / tlpk_destroy and the
 * device-resident loops (tlpk_ipm_*, tlpk_mpc_*); the device-pointer and split-phase calls belong to single and sharded handles. */
int tlpk_create_multi(tlpk_handle **out, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval,
                      const double *nzval, int index_base, const tlpk_options *opt, int ngpus, const int32_t *devices);

/* Block-angular structure of an m x n CSC matrix (int64 indices, index_base in {0,1}): row_block[i] = block id >= 0 of row i,
 * or -1 for a linking row -- the vector tlpk_options.row_block takes.  Rows are adjacent when they share a column; the densest
 * rows are removed (at most max_link_rows, 0 = max(64, m / 20)) until no connected component of the rest holds more than half of
 * the remaining rows -- or, for an LP with one dominant block, at most 80 % of them next to a second component of block size
 * (TLPK_DETECT_MAX_FRACTION) --; a small such set of linking rows is found by a geometric probe + bisection (the acceptance test is not
 * monotone in the number of removed rows, so "small", not "smallest"); small components (isolated rows) are packed into the blocks.  *n_blocks = number of diagonal blocks, 1 = no block structure (then every row_block[i] = 0 and the caller should
 * pass row_block = NULL).  Host only, deterministic, O(nnz log max_link_rows).  n_blocks / n_link may be NULL. */
int tlpk_detect_blocks(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, int index_base,
                       int64_t max_link_rows, int64_t *row_block /*m*/, int64_t *n_blocks, int64_t *n_link);

/* Introspection */
int tlpk_info(const tlpk_handle *h, tlpk_stats *out);
int tlpk_kernel_timing(const tlpk_handle *h, tlpk_kernel_times *out);
int tlpk_set_profile(tlpk_handle *h, int on);   /* toggle per-launch HIP-event timing at run time; while on, the
                                                   stream groups are serialised on the main stream so that the
                                                   per-kernel durations are not inflated by overlap */
int tlpk_get_perm(const tlpk_handle *h, int64_t *perm /*m, 0-based, perm[new] = old*/);
/* Symbolic structures, for tests and tools.  `what` selects an array; returns its length and,
 * if buf != NULL, copies min(len, cap) int64 entries. */
int64_t tlpk_symbolic_get(const tlpk_handle *h, const char *what, int64_t *buf, int64_t cap);
int64_t tlpk_symbolic_get_f64(const tlpk_handle *h, const char *what, double *buf, int64_t cap);
/* Copy the numeric factor panels (device -> host), nnzL_stored doubles.  Layout: front s (symbolic arrays front_f, front_ns, front_loff,
 * front_lda) stores its f x ns panel column-major by 64-column slices -- slice b = columns [64 b, 64 b + 64) from row 64 b down, leading
 * dimension lda - 64 b; entry (row, col) at loff + col * lda - 64 b (col - 32 b - 31) + row, b = col / 64 (fronts of <= 64 pivot
 * columns: plain column-major with leading dimension lda). */
int tlpk_get_factor(tlpk_handle *h, double *lval, int64_t cap);

/* ---------------------------------------------------------------------------------------------
 * Device-resident HSD iterate: an OPTIONAL extension for callers that keep the
 * interior-point vectors in HBM.  The drop-in interface above moves 16 (m + n) bytes over PCIe per
 * solve and leaves every right-hand side to the host; here one call runs one routine of
 * /root/reference/src/IPM/HSD/{HSD.jl, step.jl} on device vectors owned by the handle and returns only
 * scalars.  The host keeps tau, kappa, the regularisation scalars and the control flow
 * (tulip.jl_amd/hsd_device.py mirrors HSD.jl:203-350).
 * Handles: one rank, or a tlpk_create_multi handle (K1 or K2 either way).  On several devices every shard holds the sub-LP of its
 * diagonal blocks in vectors of the job's length (its columns with costs and bounds, its block rows of b, the linking rows with b
 * on the lead shard and A restricted to its columns): the same kernels then produce each shard's share of every sum / maximum /
 * minimum, the host combines them in shard order, the KKT solves run split-phase with every shard's partial xi_p on the linking
 * rows (the library's reduction of the root right-hand side completes them), and |rp|inf, |A x|inf on the linking rows come from
 * the shards' partial rows summed on the host.  The iterates equal the single-device ones up to the re-association of those
 * sums.  Sharded handles (nranks > 1) are refused: their reductions belong to the caller.
 * --------------------------------------------------------------------------------------------- */
/* b (m), c (n), l, u (n; +-Inf allowed) of the standard form (ipmdata.jl:64-173); sets the HSD starting point
 * (HSD.jl:238-247).  tlpk_ipm_reset restores the starting point. */
int tlpk_ipm_load(tlpk_handle *h, const double *b, const double *c, const double *l, const double *u);
int tlpk_ipm_reset(tlpk_handle *h);
/* HSD.jl:77-128, 136-196.  out[13] = { |rp|inf, |rl|inf, |ru|inf, |rd|inf, c'x, b'y, lz'zl, uz'zu, xl'zl + xu'zu,
 *                                      |Ax|inf, |(x-xl) lflag|inf, |(x+xu) uflag|inf, |A'y + zl lflag - zu uflag|inf } */
int tlpk_ipm_residuals(tlpk_handle *h, double tau, double *out);
/* step.jl:24-51: theta_inv from the iterate, uniform regP / regD, KKT.update!; TLPK_NOT_POSDEF -> retry with larger values */
int tlpk_ipm_factor(tlpk_handle *h, double regP, double regD);
/* step.jl:56-76: h-system; out[0] = lz'(lz th_l) + uz'(uz th_u) - (c + th_l lz + th_u uz)'hx + b'hy */
int tlpk_ipm_hsolve(tlpk_handle *h, double *out);
/* step.jl:325-364: centrality targets from the accepted direction; out[2] = { sum(vl), sum(vu) } */
int tlpk_ipm_targets(tlpk_handle *h, double a_, double mu_l, double mu_u, double *out);
/* step.jl:198-266 + 294-306: one Newton system.  mode 0 predictor | 1 corrector | 2 centrality corrector;
 * sc[8] = { tau, kappa, h0, xi_g, xi_tk, eta, gamma*mu, delta }; out[3] = { dtau, dkappa, max step to the boundary } */
int tlpk_ipm_newton(tlpk_handle *h, int mode, const double *sc, double *out);
/* step.jl:56-94: tlpk_ipm_hsolve + tlpk_ipm_newton(mode 0) with the two independent solves sharing one pass over the factor
 * (tlpk_solve2_device).  sc[8] as above except sc[2] = regG (h0 is formed inside); out[4] = { dtau, dkappa, max step, h0 } */
int tlpk_ipm_hsolve_newton(tlpk_handle *h, const double *sc, double *out);
/* step.jl:24-94: tlpk_ipm_factor without the wait for its status + tlpk_ipm_hsolve_newton: the paired solve's block-level forward sweeps
 * overlap the factorisation of the root front (tlpk_update_device_async); returns TLPK_NOT_POSDEF where tlpk_ipm_factor would have */
int tlpk_ipm_factor_hsolve_newton(tlpk_handle *h, double regP, double regD, const double *sc, double *out);
int tlpk_ipm_accept(tlpk_handle *h);                         /* step.jl:112-118: candidate -> accepted direction */
int tlpk_ipm_advance(tlpk_handle *h, double alpha, double *out);   /* step.jl:139-148; out[0] = xl'zl + xu'zu */
/* what = 0 x, 1 xl, 2 xu, 3 zl, 4 zu (n), 5 y (m) */
int tlpk_ipm_get(tlpk_handle *h, int what, double *host, int64_t len);

/* ---- Mehrotra predictor-corrector with the iterate in HBM (/root/reference/src/IPM/MPC/MPC.jl:218-410,
 * MPC/step.jl:10-358).  Same vectors as above: tlpk_ipm_load once, tlpk_ipm_residuals with tau = 1 (MPC.jl:101-141),
 * tlpk_ipm_factor (step.jl:24-51), tlpk_ipm_accept and tlpk_ipm_get are shared. */
int tlpk_mpc_start(tlpk_handle *h, double *out);               /* MPC.jl:353-410 starting point; out[0] = xl'zl + xu'zu */
/* mode 0 predictor | 1 corrector (gmu = sigma mu) | 2 centrality corrector; out[0], out[1] = largest primal / dual step
 * to the boundary of the written direction (inf if unbounded)   MPC/step.jl:164-217 */
int tlpk_mpc_newton(tlpk_handle *h, int mode, double gmu, double *out);
/* out[0] = complementarity at the point moved by (ap, ad) along the accepted direction, out[1] = at the point   step.jl:246-258 */
int tlpk_mpc_gap(tlpk_handle *h, double ap, double ad, double *out);
int tlpk_mpc_targets(tlpk_handle *h, double ap_, double ad_, double tmin, double tmax);      /* step.jl:329-358 */
int tlpk_mpc_advance(tlpk_handle *h, double ap, double ad, double *out);                     /* step.jl:112-123; out[0] = xl'zl + xu'zu */

const char *tlpk_strerror(int code);
const char *tlpk_last_error(const tlpk_handle *h);
/* message of the last FAILED tlpk_create / tlpk_create_multi of the calling thread ("" after a successful one): a failed create returns no handle */
const char *tlpk_last_create_error(void);
const char *tlpk_backend_name(void);         /* "HIP (gfx950)" */
const char *tlpk_system_name(void);          /* "Normal equations (K1)" */
const char *tlpk_linear_system(const tlpk_handle *h);   /* KKT.linear_system of this handle: "... (K1)" | "Augmented system (K2)" */
int tlpk_device_count(void);
/* host threads (pool workers + the caller) that stage the vectors of the host-pointer calls tlpk_update / tlpk_solve through pinned memory;
 * TLPK_COPY_THREADS = number of workers (default 4, 0 = the caller copies alone), read when the pool is first used */
int tlpk_host_copy_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* TLPK_H */
