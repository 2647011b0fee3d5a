#!/usr/bin/env python3
"""bench.py -- Newton-step throughput of the HIP normal-equations KKT backend on MI355X.

Metric (BASELINE.json): IPM Newton-step time = one KKT.update! (form A*D*A'+Rd, numeric
supernodal Cholesky) + `--solves` KKT.solve! calls, reported as ms per step and steps/s.
A "step" is one pass of that hot path over one synthetic input set, with theta_inv, regP, regD,
xi_p, xi_d already resident in HBM and results left in HBM (device-pointer C ABI).

Workload at N=1: BASELINE.json configs[3] -- synthetic block-angular LP, 64 blocks x
(5000 rows x 10000 vars, 4 nnz/col) + 1000 linking rows (SURVEY.md 8d's recorded choices),
the configuration the multi-GPU metric is quoted on and the largest one that fits one GPU
(configs[1]/[4] need Netlib .mps files that are not in the image; configs[2] has a ~0.86 TB
factor).  At N>1 the same LP is sharded by diagonal blocks over N ranks ("strong" scaling);
the linking-block Schur complement is all-reduced with RCCL (torch.distributed, backend nccl).

Prints ONE JSON line on rank 0, including `roofline` (dominant kernel: the fp64-MFMA panel
update, timed live with HIP events on the library's stream) and `cpu_baseline` (the C oracle on
a bounded sample, timed on this host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X datasheet fp64 matrix peak (not tabulated in the guide;
                                 # = 32 flop/clk/SIMD x 1024 SIMD x 2.4 GHz); see DESIGN.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--solves", type=int, default=4, help="KKT.solve! calls per Newton step (HSD: 3-6, MPC: 2-5)")
    ap.add_argument("--blocks", type=int, default=64)
    ap.add_argument("--mk", type=int, default=5000)
    ap.add_argument("--nk", type=int, default=10000)
    ap.add_argument("--m0", type=int, default=1000)
    ap.add_argument("--nnz-col", type=int, default=4)
    ap.add_argument("--regime", default="mid", choices=["mid", "late"])
    ap.add_argument("--workload", default="c4", choices=["c4", "headline", "c3"],
                    help="c4: BASELINE configs[3] (default). headline: the north-star instance, 100 blocks x "
                         "(2e4 inequality rows x 1e4 vars) + 1e3 linking rows = 1e6 vars / 2e6 constraints. "
                         "c3: BASELINE configs[2] (general sparse, A = [A0 I], 25 nnz/col) at --c3-rows rows "
                         "(the 5e5-row original has a ~0.86 TB factor); single GPU only")
    ap.add_argument("--c3-rows", type=int, default=50000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def cpu_baseline(args):
    """Oracle (`port`, 1 thread) on ONE diagonal block of the workload: update + `solves` solves.
    The block-angular LP is 64 such blocks plus the linking Schur complement; the reported value
    extrapolates blocks x t_block (linking work excluded, which favours the CPU)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tulip_jl_amd as tk
    from oracle_binding import OracleK1
    from workloads import block_angular_lp, kernel_inputs
    A, _ = block_angular_lp(args.blocks, args.mk, args.nk, 0, args.nnz_col, 0.5, blocks=[0], ineq=args.ineq)
    m, n = A.shape
    perm = tk.setup(A, tk.K1(), tk.Backend(device=-1)).perm()      # same fill-reducing ordering
    th, rp, rd, xp, xd = kernel_inputs(m, n, 7, args.regime)
    orc = OracleK1(A, perm)
    t0 = time.perf_counter()
    orc.update(th, rp, rd)
    for _ in range(args.solves):
        orc.solve(xp, xd)
    t = time.perf_counter() - t0
    return {"value": 1.0 / (t * args.blocks), "unit": "iter/s", "cores": 1, "kind": "port",
            "sample": f"1 of {args.blocks} diagonal blocks ({m}x{n}, nnzL={orc.nnzL}): 1 update + {args.solves} "
                      f"solves took {t:.2f} s on 1 core; value = 1/({args.blocks} x that), linking rows excluded",
            "seconds_sample": t}


def main():
    args = parse()
    args.ineq = False
    if args.workload == "headline":
        args.blocks, args.mk, args.nk, args.m0, args.ineq = 100, 20000, 10000, 1000, True
    import torch
    import tulip_jl_amd as tk
    from workloads import block_angular_lp, kernel_inputs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    if args.workload == "c3":
        if world > 1:
            raise SystemExit("general sparse LPs run on one GPU (replicas only, SURVEY.md 8e)")
        from workloads import general_sparse_lp
        A, row_block = general_sparse_lp(args.c3_rows), None
        args.no_cpu_baseline = True        # the oracle needs minutes at this size; see tests for its parity role
    else:
        A, row_block = block_angular_lp(args.blocks, args.mk, args.nk, args.m0, args.nnz_col, 0.5, ineq=args.ineq)
    m, n = A.shape
    kkt = tk.setup(A, tk.K1(), tk.Backend(device=local_rank, row_block=row_block, rank=rank, nranks=world))
    st = kkt.stats()
    th, rp, rd, xp, xd = kernel_inputs(m, n, 7, args.regime)
    dev = torch.device("cuda", local_rank)
    d_th, d_rp, d_rd, d_xp, d_xd = (torch.from_numpy(v).to(dev) for v in (th, rp, rd, xp, xd))
    d_dx = torch.empty(n, dtype=torch.float64, device=dev)
    d_dy = torch.empty(m, dtype=torch.float64, device=dev)
    P = lambda t: t.data_ptr()   # noqa: E731

    root_t = rhs_t = None
    if world > 1:                         # torch-owned buffers for the two collectives
        _, c = kkt.root_panel()
        root_t = torch.empty(c, dtype=torch.float64, device=dev) if c else None
        _, c = kkt.root_rhs()
        rhs_t = torch.empty(c, dtype=torch.float64, device=dev) if c else None

    def reduce_root(which, buf):
        if buf is None:
            return
        kkt.root_copy(which, "out", P(buf))
        kkt.sync()
        dist.all_reduce(buf)
        torch.cuda.current_stream().synchronize()
        kkt.root_copy(which, "in", P(buf))

    def newton_step():
        if world == 1:
            kkt.update_device(P(d_th), P(d_rp), P(d_rd))
            for _ in range(args.solves):
                kkt.solve_device(P(d_dx), P(d_dy), P(d_xp), P(d_xd), sync=False)
            kkt.sync()
        else:
            kkt.update_local(P(d_th), P(d_rp), P(d_rd))
            reduce_root("panel", root_t)
            kkt.update_finish()
            for _ in range(args.solves):
                kkt.solve_local(P(d_xp), P(d_xd))
                reduce_root("rhs", rhs_t)
                kkt.solve_finish(P(d_dx), P(d_dy), P(d_xd))
            kkt.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        newton_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        newton_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps

    # sanity: the residual identities of the reference's conformance test on the last step
    dx, dy = d_dx.cpu().numpy(), d_dy.cpu().numpy()
    if dist is not None:       # block rows / columns live on their owner; linking rows replicated
        t1, t2 = d_dx.clone(), d_dy.clone()
        dist.all_reduce(t1)
        lk = torch.from_numpy((row_block < 0)).to(dev)   # world > 1 implies a block-angular workload
        t2 = torch.where(lk, t2 / world, t2)
        dist.all_reduce(t2)
        dx, dy = t1.cpu().numpy(), t2.cpu().numpy()
    r_p = float(np.abs(A @ dx + rd * dy - xp).max())
    r_d = float(np.abs(-dx * (th + rp) + A.T @ dy - xd).max())

    out = {
        "metric": "IPM Newton-step rate: KKT.update! (A*D*A'+Rd, supernodal Cholesky) + %d KKT.solve!" % args.solves,
        "value": 1e3 / ms_per_step, "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[2] at reduced scale: general sparse LP A=[A0 I], %d rows x %d structural "
                                "columns, 25 nnz/col; m=%d n=%d nnz(A)=%d" % (m, n - m, m, n, A.nnz)) if args.workload == "c3" else
                               "%s: block-angular LP, %d blocks x (%d %s rows x %d vars, %d nnz/col) + %d linking rows; "
                               "m=%d n=%d nnz(A)=%d" % ("BASELINE configs[3]" if args.workload == "c4" else "north-star headline",
                                                        args.blocks, args.mk, "inequality" if args.ineq else "equality",
                                                        args.nk, args.nnz_col, args.m0, m, n, A.nnz),
                   "solves_per_step": args.solves, "regime": args.regime, "parallelism": "blocks/%d" % world,
                   "stream_groups": int(kkt.symbolic("ngroups")[0]),
                   "nnzS": st["nnzS"], "nnzL": st["nnzL"], "nnzL_stored": st["nnzL_stored"],
                   "flops_chol": st["flops_chol"], "n_supernodes": st["n_supernodes"], "n_levels": st["n_levels"],
                   "max_front": st["max_front"], "launches_update": st["launches_update"],
                   "launches_solve": st["launches_solve"], "ms_analyse": st["ms_analyse"],
                   "residual_inf": [r_p, r_d]},
    }

    if not args.no_roofline:
        # Per-kernel-class device time of one Newton step, HIP events around every launch on the
        # stream it is launched on.  The timed region above runs the diagonal blocks on concurrent
        # stream groups (plus side streams); under that overlap a kernel's [start, end] interval includes
        # time it shares the chip with other groups' kernels, so the roofline leg replays the SAME
        # LP and the SAME kernels with a single-stream schedule (streams=1), where every launch
        # has the device to itself.  profiles/*kernel_stats.csv is taken the same way.
        del d_dx  # free a little before the second handle
        d_dx = torch.empty(n, dtype=torch.float64, device=dev)
        kkt1 = kkt if st["n_blocks"] < 2 else tk.setup(
            A, tk.K1(), tk.Backend(device=local_rank, row_block=row_block, rank=rank, nranks=world, streams=1))
        main_kkt, kkt = kkt, kkt1
        newton_step()                                   # warm-up of the second handle
        kkt.set_profile(True)
        newton_step()
        kt = kkt.kernel_times()
        kkt.set_profile(False)
        upd = kt["update"]
        fl = kkt.stats()["flops_update"]
        kkt = main_kkt
        ach = fl / (upd["ms"] * 1e-3) / 1e12 if upd["ms"] > 0 else 0.0
        traffic = None
        try:        # HBM bytes per launch from the committed PMC pass of this command (cannot be collected in-process)
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_k_update.json")))
            if pm.get("workload") == args.workload and world == 1:
                traffic = pm["traffic_bytes_per_launch"]
        except Exception:
            pass
        out["roofline"] = {"bound": "mfma", "kernel": "k_update (v_mfma_f64_16x16x4_f64)", "achieved": ach,
                           "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                           "traffic": traffic, "launches": upd["launches"],
                           "avg_launch_ms": upd["ms"] / max(upd["launches"], 1), "flops_per_step": fl,
                           "flops_per_launch": fl / max(upd["launches"], 1),
                           "peak_measured": 77.9, "frac_measured": ach / 77.9, "measured_with_streams": 1}
        solve_bytes = 2 * 8 * st["nnzL"] + 2 * 12 * A.nnz + 8 * (4 * n + 3 * m)
        sol_ms = (kt["solve_fwd"]["ms"] + kt["solve_bwd"]["ms"] + kt["spmv"]["ms"]) / max(args.solves, 1)
        out["kernel_ms"] = {k: round(v["ms"], 4) for k, v in kt.items()}
        out["solve_roofline"] = {"bound": "hbm", "achieved": solve_bytes / (sol_ms * 1e-3) / 1e9 if sol_ms > 0 else 0.0,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_solve": solve_bytes,
                                 "ms_per_solve": sol_ms}
        out["solve_roofline"]["frac"] = out["solve_roofline"]["achieved"] / HBM_PEAK_GBS
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
