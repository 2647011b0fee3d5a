// tlpk_handle.hpp -- the solver object behind `tlpk_handle*` and the small helpers shared by the
// translation units of the C ABI (tlpk_api.cpp: KKT interface; tlpk_ipm.cpp: device-resident IPM vectors).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <array>
#include <vector>

#include "../../include/tlpk.h"
#include "tlpk_device.hpp"

using namespace tlpk;

struct IpmState;        // tlpk_ipm.cpp
constexpr int MAX_DEVICES = 16;

struct tlpk_handle {
    Symbolic S;
    Options opt;
    std::vector<i64> row_block_copy, user_perm_copy;
    int device = -1;
    bool has_device = false;
    bool profile = false;
    int fault_at = -1, n_updates = 0; bool fault_done = false;      // TLPK_CHAIN_FAULT (testing): see enq_update_local
    int chain_retries = 0;        // tlpk_update: replays after a dependency-driven launch gave up waiting
    bool shared_device = false;   // set by tlpk_create_multi when another shard of the job names the same device (Options::shared_device)
    bool serial = false;          // TLPK_SERIAL=1: every launch on the main stream (what profile mode does), for external profilers
    hipStream_t stream = nullptr;                 // main stream (= group 0)
    hipStream_t gstream[MAX_GROUPS] = {};         // gstream[0] == stream; others: concurrent subtree groups
    hipEvent_t ev_fork = nullptr, ev_join[MAX_GROUPS] = {};
    hipStream_t zstream = nullptr;        // lowest priority: the upper fronts' zero-fill + assembly must not take slots from the leaf levels' launches
    hipEvent_t ev_zfork = nullptr, ev_upper = nullptr;   // zero-fill + assembly of the upper fronts beside the leaf levels (symbolic.cpp step 13d)
    bool upper_split = false;             // this update runs them on the last group's side stream (set per call: not in the single-stream modes / graphs)
    hipStream_t sstream[MAX_GROUPS] = {};         // side stream of each group: diagonal-block chains overlap the bulk update
    hipEvent_t ev_side[MAX_GROUPS] = {};
    bool forked = false;
    hipStream_t rstream = nullptr;                // the root (linking) front of tlpk_update_device_async runs here, beside the next solve's block sweeps
    hipEvent_t ev_blocks = nullptr, ev_root = nullptr;
    bool root_pending = false;                    // an asynchronous update has not been waited for / checked yet
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_class;            // class of each recorded pair in the current call
    std::vector<std::array<long long, 3>> ev_launch;   // (kind, first, count) of the launch behind each pair (-1: not a schedule launch): TLPK_PROF_DUMP
    size_t ev_used = 0;
    DevArrays d;
    std::vector<void *> allocs;
    i64 device_bytes = 0;
    double *d_theta = nullptr, *d_regP = nullptr, *d_regD = nullptr, *d_D = nullptr;
    double *d_xip = nullptr, *d_xid = nullptr, *d_dx = nullptr, *d_dy = nullptr;
    int refine_steps = 0;                         // tlpk_options.refine_steps
    double *d_r1 = nullptr, *d_r2 = nullptr, *d_cx = nullptr, *d_cy = nullptr;   // refinement: residuals and correction
    unsigned long long *d_ref = nullptr;          // guarded refinement: norms and verdict words (kernels.hip: k_refine_decide)
    i64 refine_rejected = 0;                      // refinement steps of the last solve that did not shrink the residual and were discarded
    double *d_bx = nullptr, *d_by = nullptr;      // multi-device refinement: the iterate before a step (restored if the step is rejected)
    int *h_info = nullptr;
    double *pin_in = nullptr, *pin_out = nullptr;   // pinned staging of the host-pointer entry points (lazily allocated)
    bool io_timing = false; double io_t_first = 0, io_t_last = 0;   // TLPK_HOSTIO_TIMING: when the first / last device-to-host group had landed
    std::vector<hipEvent_t> io_events;              // one per device-to-host piece of tlpk_solve (tlpk_api.cpp: stage_out)
    bool factored = false, local_done = false, solve_local_done = false, solve_timed = false, refine_pending = false, pair_pending = false;
    i64 fail_col = -1;
    double ms_analyse = 0, ms_update = 0, ms_solve = 0;
    tlpk_kernel_times kt{};
    size_t factor_marker = 0, fwd_marker = 0;   // index of the LK_ALLREDUCE_ROOT launch (or size)
    i64 first_link = 0, nlink = 0;
    // persistent sweeps: ticket-counter slot of every sweep launch
    std::vector<i32> sweep_slot_fwd, sweep_slot_bwd;
    unsigned long long solve_epoch = 0;
    int poll[3] = {8, 16, 32};          // TLPK_POLL=fast,nfast,slow (tuning knob of the sweep kernels' polling back-off)
    // single-process multi-device mode (tlpk_create_multi): the parent owns one sharded handle per device
    std::vector<tlpk_handle *> sub;
    double *multi_tmp = nullptr;        // device 0: staging of the peers' root panels / root rhs for the reduction
    i64 multi_red_off = 0, multi_dy0_off = 0;   // ... followed by the reduced buffer and the lead's rank-local dy
    hipEvent_t multi_ev[MAX_DEVICES] = {}; hipEvent_t multi_done = nullptr;
    void *multi_comm[MAX_DEVICES] = {}; bool multi_rccl = false;   // TLPK_MULTI_REDUCE=rccl: one ncclComm_t per shard
    int multi_mode = 1;                 // parent: reduction of the root panel / rhs: 0 gather to the lead + broadcast, 1 reduce-scatter + all-gather over peer copies (default), 2 RCCL
    hipEvent_t multi_ev2[MAX_DEVICES] = {}; i64 rs_slice = 0;     // parent: 'slice broadcast' events, slice length (doubles) of the largest reduced buffer
    double *rs_stage = nullptr;         // child: staging of the peers' contributions to THIS shard's slice ((N - 1) x rs_slice doubles)
    void *shard_pool = nullptr;         // parent: one persistent host thread per shard (tlpk_api.cpp: ShardPool) -- the shards' launches are enqueued concurrently
    double ms_enqueue_update = 0;       // parent: host time from the entry of tlpk_update until every shard's work is enqueued
    i64 col_lo = 0, col_hi = 0, row_lo = 0, row_hi = 0, link_lo = 0, link_hi = 0;   // child: slices of the job-wide input vectors it reads
    double *shared_dy = nullptr;        // child: job-wide dy on the lead device (P2P), filled with the rows this rank owns
    bool dx_local_only = false;         // child: dx is the job-wide vector, leave the other ranks' columns alone
    bool stagger = false, stagger_armed = false; i64 stagger_min = 10000; hipEvent_t ev_stagger = nullptr;   // TLPK_STAGGER (experiment, tlpk_api.cpp: run_launches)
    bool rhs_all_ranks = false;         // child, device-resident IPM: this solve adds the shard's xi_p on the linking rows whatever its rank
    // hipGraph replay of the static schedules (tlpk_api.cpp: graph_or_direct): instantiated graphs and their keys
    bool use_graph = true;              // TLPK_GRAPH=0 turns it off; switched off for good if capture fails on this system
    bool force_graph = false;           // TLPK_GRAPH=2: also for schedules with concurrent stream groups
    bool update_whole = false, solve_whole = false;   // internal: the composed entry point enqueues both halves itself
    std::vector<hipGraphExec_t> graph_execs;
    std::vector<std::vector<char>> graph_keys;
    IpmState *ipm = nullptr;            // device-resident interior-point vectors (tlpk_ipm_load), freed by tlpk_destroy
    std::string last_error;
};

void ipm_free(tlpk_handle *h);          // tlpk_ipm.cpp
// tlpk_api.cpp, for the device-resident interior-point loops on a multi-device handle: KKT.update! from the theta / regP / regD every
// shard holds on its device, and KKT.solve! with shard-resident vectors in the rank-local layout (every shard adds ITS xi_p on the
// linking rows -- partial residuals --, the library's reduction completes them; nothing is gathered)
int multi_update_resident(tlpk_handle *h);
int multi_solve_resident(tlpk_handle *h, double *const *dx, double *const *dy, const double *const *xip, const double *const *xid);
int multi_solve2_resident(tlpk_handle *h, double *const *dx0, double *const *dy0, const double *const *xip0, const double *const *xid0,
                          double *const *dx1, double *const *dy1, const double *const *xip1, const double *const *xid1);



inline int hip_fail(tlpk_handle *h, hipError_t e, const char *what) {
    h->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return (e == hipErrorOutOfMemory) ? TLPK_OOM : TLPK_HIPERR;
}
#ifndef HIPCHK
#define HIPCHK(h, call)                                                   \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return hip_fail((h), e_, #call);            \
    } while (0)
#endif

template <class T>
inline int dev_alloc(tlpk_handle *h, T **out, i64 count) {
    *out = nullptr;
    const size_t bytes = (size_t)std::max<i64>(count, 1) * sizeof(T);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return hip_fail(h, e, "hipMalloc");
    h->allocs.push_back(p);
    h->device_bytes += (i64)bytes;
    *out = (T *)p;
    return TLPK_OK;
}
template <class T, class A>
inline int dev_upload(tlpk_handle *h, T **out, const std::vector<T, A> &v) {
    int rc = dev_alloc(h, out, (i64)v.size());
    if (rc != TLPK_OK) return rc;
    if (!v.empty()) HIPCHK(h, hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return TLPK_OK;
}

