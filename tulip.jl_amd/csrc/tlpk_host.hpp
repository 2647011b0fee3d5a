// tlpk_host.hpp -- host-side analyse phase of libtlpk (ordering, elimination tree, supernodes,
// assembly maps, launch schedules).  Everything here runs once per KKT.setup
// (/root/reference/src/KKT/Cholmod/spd.jl:5-20 does the same work through CHOLMOD's analyse) and
// is amortised over the IPM iterations; the numeric work is in kernels.hip.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace tlpk {

using i32 = int32_t;
using i64 = int64_t;

// ---- tunables of the dense front kernels (shared by the scheduler and kernels.hip) ----
constexpr int NB_IN = 64;     // diagonal-block / triangular-solve width (potrf + trsm kernels)
constexpr int NB_OUT = 256;   // outer panel width: trailing updates run with K = NB_OUT
constexpr int TILE = 128;     // update-kernel tile (TILE x TILE per workgroup)
constexpr int TRSM_ROWS = 64;  // rows per trsm_rows call inside the diagonal block (4 waves x 16 rows)
constexpr int SMALL_NS = 16;       // fronts with at most this many pivot columns and ...
constexpr int SMALL_ROWS = 256;    // ... at most this many rows below them take the wave-per-front solve kernels
constexpr int TRSM_THIN_W = 32;    // widest block column handled by k_trsm_thin (one thread per row, 256 rows per workgroup)
constexpr int TRSM_WG_ROWS = 64;   // rows per k_trsm workgroup (4 waves x 16 rows x whole block column)
constexpr int EA_COLS = 16;   // parent columns per extend-add workgroup
constexpr int SOLVE_NB = 128; // block width of the triangular-solve kernels (two NB_IN sub-blocks)
constexpr int SWEEP_NB = NB_IN;  // width of a pivot block handed from workgroup to workgroup in the persistent sweeps
constexpr int MAX_GROUPS = 8;    // concurrent streams for independent diagonal blocks
constexpr int LDA_PAD_MIN_F = 64;   // fronts with at least this many rows get a line-aligned panel (lda multiple of 16)
constexpr int SOLVE_ROWS = 256; // rows per forward-update workgroup
constexpr int GATHER_WIDE_PER_ROW = 16;  // forward gather: fronts whose rows collect at least this many entries on average take 8 lanes per row (32-row tasks)
constexpr int BWD_ROWS = 128;   // rows per backward-update workgroup (partial sums, reduced in fixed order)

// ---- packed panels --------------------------------------------------------------------------------------------------------
// The panel of a front (f rows, ns pivot columns, leading dimension lda) is stored by 64-column SLICES: slice b = columns
// [64 b, 64 b + 64) keeps the rows from its first row 64 b down to lda, with its own leading dimension lda - 64 b.  The blocks above the
// 64 x 64 diagonal blocks -- 39 % of a square pivot block -- are never read by any kernel and are not stored (round 3: stored / nnz(L)
// 1.82 -> 1.14 on config C4).  Entry (row, col), row >= 64 (col / 64), sits at
//     loff + pk_off(lda, col) + row,      pk_off(lda, col) = col * lda - 64 b (col - 32 b - 31),  b = col / 64
// (pk_off = the offset of the column's VIRTUAL row 0; inside one slice consecutive columns are lda - 64 b apart), and a whole panel takes
// pk_len(lda, ns) = pk_off(lda, ns) + 64 (ns / 64) doubles.  Fronts of at most 64 pivot columns are laid out as before.
#if defined(__HIPCC__)
#define TLPK_HD __host__ __device__
#else
#define TLPK_HD
#endif
TLPK_HD inline i64 pk_off(i32 lda, i32 col) { const i64 b = col >> 6; return (i64)col * lda - 64 * b * ((i64)col - 32 * b - 31); }
TLPK_HD inline i64 pk_len(i32 lda, i32 ns) { return pk_off(lda, ns) + 64 * (i64)(ns >> 6); }

// ---- device-visible descriptors (plain structs, uploaded as arrays) ----
struct FrontDesc {
    i64 loff;      // offset of the panel (f x ns, column-major in 64-column slices, see pk_off) in Lval
    i64 uoff;      // offset of the update matrix (rs x rs, ld = rs) in its ping-pong buffer
    i64 rowoff;    // offset into rowidx (f entries, first ns are the pivot columns)
    i64 reloff;    // offset into rel (rs entries: position of each below-row in the parent front)
    i64 ucoff;     // offset of the rs-vector of solve contributions
    i64 dinvoff;   // offset of the inverted diagonal blocks: block b (NB_IN columns) at dinvoff + b*NB_IN*NB_IN, ld = its nb
    i32 f, ns;
    i32 col0;      // first pivot column (permuted numbering)
    i32 ubuf;      // which ping-pong buffer holds U (depth & 1)
    i32 parent;    // parent front or -1
    i32 child_ptr, nchild;   // children in `children[child_ptr .. child_ptr+nchild)`
    i32 flagoff;   // >= 0: the front's triangular solves run in the persistent sweep kernels; -1: small / single / not local
    i32 lda;       // leading dimension of the panel: f rounded up to 16 doubles (one 128-byte line) for fronts of >= LDA_PAD_MIN_F
                   // rows, so that every panel column starts on a cache-line boundary (loff is then a multiple of 16 too)
    i32 eatab;     // offset into ea_tab: for every parent-column boundary of the parent's extend-add tasks, the first of this
                   // front's update-matrix columns that lands at or after it (-1: no parent)
};
static_assert(sizeof(FrontDesc) == 88, "FrontDesc layout");

struct PotrfTask { i32 front, k0, nb, kprev; };              // diagonal block of a block column: columns [k0, k0 + nb), nb <= NB_OUT; kprev = k0
struct TrsmTask  { i32 front, k0, nb, row0, kprev, fuse_nb, pad1, pad2; };   // rows [row0, pad1) below the diagonal block [k0, k0 + nb) of a block column; kprev = k0, fuse_nb = 0 (unused)
// seg != 0: the tile skips the K slabs (16 columns) that are structurally zero in one of its two operand row ranges (padding of
// amalgamated supernodes): upd_seg[seg - 1] = number of K segments, then (first column, slabs) per segment, all inside the full
// slabs of [k0, k0 + kw); nsl = slabs to execute (>= 2); a partial last slab (kw % 16 columns) is always executed
struct alignas(16) UpdateTask { i32 front, k0, kw, i0, j0, jlim, beta0, pad1, seg, nsl, pad2 = 0, pad3 = 0; };   // 48 bytes: scalar loads (a 40-byte struct was copied through scratch)
 // pad1 = slot + 1: split-K part, the raw tile goes to scratch slot `slot`;
// in reduce_tasks: k0 = first slot, kw = number of parts
//  // tile rows i0.., cols j0..<jlim; beta0: U targets are written, not accumulated
// front assembly (k_front_assemble): the tile [rows of boundaries br0 .. br1) x [columns of boundary bc, bc + 1) of the panel of a large front is
// FORMED in LDS -- entries of S, then the children's update matrices in child order -- and written once (no zero-fill, no read-modify-write)
struct FaTask    { i32 front, bc, br0, br1; };
constexpr int FA_CW = 4;       // parent columns per tile = the extend-add column range of the large fronts (round 5: 4 x 2304 tall tiles; round 4: 16 x 256)
constexpr int FA_RB = 576;     // row boundaries per tile: <= FA_RB * FA_CW = 2304 rows (73.7 KB of LDS per tile: two workgroups per CU)
struct EaTask    { i32 front, j0, j1, bidx, br0, br1, pad0, pad1; };   // rows of the boundaries [br0, br1) only (br1 = 0: all rows): row bands shrink a workgroup's working set of parent lines (TLPK_EA_BANDS)                       // parent columns [j0, j1) = boundaries bidx, bidx + 1 of the front's extend-add ranges
struct SolveTask { i32 front, k0, nb, row0, slot, nslot, pad0, pad1; };
// sweep items (LK_FWD_SWEEP): k0/nb = first row / rows of the chunk (<= SWEEP_NB pivot rows or <= SOLVE_NB rows below),
//   slot = 1 pivot block (solve + publish) | 0 rows below, nslot = SWEEP_NB-wide solved blocks to consume (0 .. nslot-1);
//   (LK_BWD_SWEEP): k0/nb = column block (<= SWEEP_NB), row0/slot = first row / number of rows below the pivot block
//   (values of the ancestors), nslot = later blocks to consume (descending from the last)  // forward: nslot = width of the next diagonal block solved by this workgroup (0 = none);
// backward: k0/nb = target column block, row0/slot = first source row / number of source rows, nslot != 0 = also solve the diagonal block

enum LaunchKind : i32 {
    LK_EXTEND_ADD = 0, LK_POTRF, LK_TRSM, LK_UPDATE,
    LK_FWD_GATHER, LK_FWD_DIAG, LK_FWD_UPDATE, LK_BWD_UPDATE, LK_BWD_DIAG /* unused: the diagonal solve is fused into LK_BWD_UPDATE */,
    LK_ALLREDUCE_ROOT,  // marker: everything after this belongs to the replicated root front
    LK_POTRF_WIDE,      // diagonal block wider than NB_IN (several 64-wide steps in one workgroup)
    LK_SIDE_FORK,       // marker: the group's side stream waits for the group's stream
    LK_SIDE_JOIN,       // marker: the group's stream waits for its side stream
    LK_UPDATE_REDUCE,   // applies the split-K partial tiles of the preceding LK_UPDATE launch to their targets
    LK_TRSM_THIN,       // block columns of <= TRSM_THIN_W columns: one thread per row (no MFMA strips)
    LK_FWD_SMALL, LK_BWD_SMALL,  // whole fronts of <= SMALL_NS pivot columns: one wave per front and sweep
    LK_POTRF_SMALL,     // pivot blocks of fronts with <= SMALL_NS pivot columns: one wave per front
    LK_FWD_SWEEP,       // whole forward substitution of every (non-small) front of a level in ONE launch: workgroups
    LK_BWD_SWEEP,       // own row chunks / column blocks and hand solved blocks over through flags (k_fwd_sweep / k_bwd_sweep)
    LK_FRONT_ASSEMBLE,  // panels of the large fronts of a level formed tile by tile: S entries + children (k_front_assemble)
    LK_WAIT_UPPER,      // marker: from here on the group's stream touches panels of the UPPER fronts (front_upper): wait for their zero-fill + assembly
    LK_CHAIN            // round 6: the whole blocked factorisation of a level's multi-block-column fronts in ONE persistent launch (k_chain): update tiles,
                        // diagonal blocks, triangular solves and split-K reductions are ITEMS drawn from a ticket counter, released by completion counters
    , LK_UPDATE_T64     // round 6: the LAST partial round of the preceding LK_UPDATE launch as 64 x 64 tiles (UpdateTask.pad2 = 1; k_update64: 4 waves per workgroup)
};
// ---- dependency-driven factorisation (LK_CHAIN, symbolic.cpp: build_schedule / kernels.hip: k_chain) ----
// An item is one task of the ordinary kernels (an update tile, the diagonal block of a block column, a 64-row strip of a triangular solve, one eighth of a
// split-K reduction).  It starts when the counters it names have reached their values -- every one of them is raised by items with SMALLER tickets, so a
// workgroup only ever waits for workgroups that already run -- and raises ONE counter when its stores are published (agent-scope release).
//   wait 0 / wait 1: every counter of [w, w + n) >= need   (the hand-over flags of the two operand row ranges of an update tile; the target tiles of a
//                    diagonal block / of a strip's tile row)
//   wait 2:          counter w2 >= need2                   (the earlier adder of the same target tile; the diagonal block of a strip; the parts of a reduction)
enum ChainRole : i32 { CR_UPDATE = 0, CR_POTRF = 1, CR_TRSM = 2, CR_REDUCE = 3 };
struct alignas(16) ChainItem { i32 role, task, sub, w0, n0, need0, w1, n1, need1, w2, need2, sig; };     // sub: part of a reduction (0 .. RED_SPLIT - 1); sig < 0: none
static_assert(sizeof(ChainItem) == 48, "ChainItem layout");
constexpr int RED_SPLIT = 8;       // workgroups (items) per split-K reduction tile
struct Launch { i32 kind; i32 group; i64 first; i64 count; i32 side = 0; i32 pad = 0; };   // tasks[first .. first+count); group: stream (-1 = after all groups joined)

struct Options {
    i32 ordering = 0, relax = 1, rank = 0, nranks = 1, streams = 0;
    i32 system = 0;                   // 0 = K1 normal equations, 1 = K2 augmented system (signed Cholesky)
    i64 k2_n = 0;                     // K2, internal: number of variable nodes (nodes [0, k2_n) are variables, the rest constraints)
    i32 analyse_div = 0;              // host threads of the analyse phase = default / analyse_div; 0 = nranks (N ranks analyse at the same time on one host); tlpk_create_multi: 1 for
                                      // its ONE rank-independent analysis
    i32 shared_device = 0;            // another shard of the same job runs on this shard's device (tlpk_create_multi with a device named twice: test configurations).  The
                                      // strips of the dependency-driven launches then keep their plain waits (no early entry, see build_chain): with eight shards' launches on one
                                      // GPU the in-role waits gave up in ~1 % of the runs of the eight-shards tests even with lean polling (profiles/r06_chain_poll_storm.txt)
    const i64 *user_perm = nullptr;   // 0-based here
    const i64 *row_block = nullptr;
};

// std::vector without the value-initialisation of resize(n): the large arrays of the analyse phase are written exactly once, by the host threads that own their
// parts -- a serial zero-fill in front of that costs a page fault per 4 KB on one thread (round 5: a third of the "assembly lists" phase).
template <class T>
struct NoInit : std::allocator<T> {
    template <class U> struct rebind { using other = NoInit<U>; };
    NoInit() = default;
    template <class U> NoInit(const NoInit<U> &) {}
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void *>(p)) U; else ::new (static_cast<void *>(p)) U(std::forward<A>(a)...);
    }
};
template <class T> using uvec = std::vector<T, NoInit<T>>;

struct Symbolic {
    i64 m = 0, n = 0, nnzA = 0;          // K2: m = order of the augmented matrix (n_var + m_con), n / nnzA those of the incidence matrix below
    i32 system = 0; i64 k2_n = 0, k2_m = 0;   // K2: user dimensions (variables, constraints)
    i32 shared_device = 0;               // Options::shared_device of the rank this schedule is built for
    std::vector<double> csign;           // K2: +1 / -1 per permuted column (constraint / variable node)
    // A, CSC and CSR (0-based, int32 indices); csr_pos[q] = CSC position of the CSR entry q
    std::vector<i64> Ap; std::vector<i32> Ai; std::vector<double> Ax;
    std::vector<i64> Tp; std::vector<i32> Tj; std::vector<i32> Tpos;
    std::vector<i32> Acol;                 // column of each CSC entry
    // ordering
    std::vector<i32> perm, iperm;          // perm[new] = old
    // permuted lower-triangular pattern of S (diagonal first in each column)
    std::vector<i64> Sp; std::vector<i32> Si;
    std::vector<i32> parent;               // column elimination tree (postordered numbering)
    std::vector<i32> colcount;             // nnz(L[:,j]) incl. diagonal
    // supernodes / fronts
    i32 nsuper = 0;
    std::vector<FrontDesc> fronts;
    std::vector<i32> sn_of_col;
    std::vector<i32> rowidx;               // concatenated front row lists (permuted indices)
    std::vector<i32> rel;                  // concatenated relative indices
    std::vector<i32> ea_tab;               // per child front: lower bounds of its rel list at the parent's extend-add range boundaries
    std::vector<i64> gth_ptr, gth_src;     // forward gather lists: front row (rowoff + t) -> children's uc entries, in child order
    std::vector<i32> children;             // concatenated child lists
    std::vector<i32> depth;                // depth of each front (roots = 0)
    i32 nlevels = 0;
    std::vector<i32> level_ptr, level_fronts;   // level d: level_fronts[level_ptr[d]..level_ptr[d+1])
    // block-angular structure
    i32 nblocks = 0;
    std::vector<i32> front_block;          // block of each front, -1 for the root/linking front
    std::vector<char> front_local;         // processed by this rank
    std::vector<char> front_single;        // isolated 1 x 1 front (f = ns = 1, no children): handled by k_single_*, not by the schedules
    std::vector<i64> single_loff, single_dinvoff; std::vector<i32> single_col;   // the local ones, for the device
    std::vector<i32> zero_tasks;           // (front, first column) of every 64-column slice of a local panel: k_zero_panels
    std::vector<i32> zero_small;           // local fronts whose whole panel (<= 4096 entries) one wave zeroes
    std::vector<char> front_upper;         // 1 = big front of the top levels: its zero-fill + assembly run on a stream of their own beside the leaf levels (step 13d)
    i64 n_zero_lower = 0;                  // zero_tasks: the first n_zero_lower (front, c0) pairs belong to the other ("lower") fronts
    std::vector<char> col_local;           // column of A handled by this rank
    std::vector<char> row_local;           // 0 = other rank's block row, 1 = local block row, 2 = linking row
    i32 root_front = -1;                   // the replicated linking front (or -1)
    i32 ngroups = 1;                       // independent subtree groups run on concurrent streams
    std::vector<i32> front_group;          // group of each front (fronts at depth 0 run after the join)
    i32 n_local_blocks = 0;
    // assembly of S = A*D*A' + Rd into the panels
    uvec<i64> s_target;                    // per S entry: position in Lval
    uvec<i32> s_diag_row;                  // per S entry: original row index if diagonal, else -1
    uvec<i64> pair_ptr;                    // per S entry: range of products
    uvec<double> pair_w;                   // A[i,j]*A[k,j]
    uvec<i32> pair_j;                      // j
    uvec<char> s_local;                    // entry assembled by this rank
    // sizes
    i64 nnzS = 0, nnzL = 0, lval_len = 0, ubuf_len[2] = {0, 0}, uc_len = 0, max_front = 0, dinv_len = 0;
    double flops_chol = 0, flops_panel = 0, flops_update = 0, flops_update_alg = 0;
    double flops_update_chain = 0, flops_update_alg_chain = 0;   // the share of the two that runs inside the LK_CHAIN launches (k_chain), not in k_update
    // schedules
    std::vector<PotrfTask> potrf_tasks; std::vector<TrsmTask> trsm_tasks;
    std::vector<UpdateTask> update_tasks, reduce_tasks; std::vector<EaTask> ea_tasks;
    std::vector<FaTask> fa_tasks;          // tiles of the panels formed by k_front_assemble
    std::vector<char> front_fa;            // front whose panel is formed by k_front_assemble (no zero-fill, not in k_assemble, no panel-part extend-add)
    std::vector<i32> upd_seg;              // K-segment lists of the update tasks that skip structurally zero slabs (UpdateTask.seg)
    // structural-zero flags of the amalgamated fronts (analyse_rank step 13c; host only): skip_off[s] = -1, or the offset in skip_bits of
    // front s: one bit per (16-column K slab, 16-row group), slab-major, skip_words(s) 64-bit words per slab
    std::vector<i64> skip_off; std::vector<uint64_t> skip_bits;
    double flops_update_skipped = 0;       // padded flops the skip lists leave out (flops_update counts what is executed)
    i64 spart_len = 0;                     // split-K scratch: TILE x TILE doubles per partial tile
    std::vector<SolveTask> fwd_gather_tasks, fwd_diag_tasks, fwd_update_tasks, bwd_update_tasks, fwd_small_tasks, bwd_small_tasks;
    std::vector<SolveTask> fwd_sweep_tasks, bwd_sweep_tasks;
    i64 n_sweep_flags = 0;                 // number of fronts handled by the sweep kernels
    bool sweep = true;                     // persistent sweep kernels (TLPK_SWEEP=0: one launch per 128-column block step)
    bool solve_single_stream = true;       // every launch of the solve schedules runs on the handle's main stream (one stream group, or solve_one_group)
    std::vector<Launch> factor_launches, fwd_launches, bwd_launches;
    std::vector<ChainItem> chain_items;    // items of the LK_CHAIN launches (Launch.first / count index this list; Launch.pad = counter index of the launch's ticket)
    i64 chain_counters = 0;                // completion counters + tickets of all LK_CHAIN launches (u32 each, zeroed at the start of every update!)
    // state handed from analyse_common to analyse_rank (rank-independent)
    std::vector<i32> row_block_v, col_block_v;   // block of each row / column of A (-1: linking), empty = general sparse
    std::vector<i32> sparent_v;                  // parent of each front
    i32 nlink_v = 0;                             // number of linking rows
    std::string error;
};

// amd.cpp
void amd_order(i32 n, const std::vector<i64> &xadj, const std::vector<i32> &adj, std::vector<i32> &order);

// symbolic.cpp : returns a TLPK_* code.  analyse_k2 builds the structures of the augmented system
// [-(Theta^-1 + Rp) A'; A Rd] (order n + m) from the same machinery.
int analyse(Symbolic &S, i64 m, i64 n, const i64 *colptr, const i64 *rowval, const double *nzval,
            int index_base, const Options &opt);
// the two halves of analyse(): the rank-independent part (ordering ... front structures) and one rank's ownership, storage
// offsets, lists and schedules; a copy of a Symbolic after analyse_common can be finished for any (rank, nranks)
int analyse_common(Symbolic &S, i64 m, i64 n, const i64 *colptr, const i64 *rowval, const double *nzval,
                   int index_base, const Options &opt);
int analyse_rank(Symbolic &S, const Options &opt);
// K2: analyse_k2 = analyse_k2_common (incidence matrix of the augmented system, rank-independent analysis, signs; *opt_out = the
// options analyse_rank needs: system, k2_n) + analyse_rank
int analyse_k2_common(Symbolic &S, i64 m, i64 n, const i64 *colptr, const i64 *rowval, const double *nzval,
                      int index_base, const Options &opt, Options *opt_out);
int analyse_k2(Symbolic &S, i64 m, i64 n, const i64 *colptr, const i64 *rowval, const double *nzval,
               int index_base, const Options &opt);

}  // namespace tlpk
