#!/usr/bin/env python3
"""Per-rank device time of one Newton step of config C4 when the LP is sharded over N ranks, measured
on ONE GPU: rank 0 of N runs its local half-steps (update_local/update_finish, solve_local/
solve_finish) without the all-reduces (results are not meaningful, the timing of the local work is).
Used to size the strong-scaling expectation of bench.py --gpus N on a single-GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tulip_jl_amd as tk
from workloads import block_angular_lp, kernel_inputs

HEADLINE = os.environ.get("HEADLINE") == "1"      # the north-star instance (100 blocks x (20 000 inequality rows x 10 000 vars) + 1000 linking rows)
A, row_block = block_angular_lp(100, 20000, 10000, 1000, 4, 0.5, ineq=True) if HEADLINE else block_angular_lp()
m, n = A.shape
th, rp, rd, xp, xd = kernel_inputs(m, n, 7, "mid")
dev = torch.device("cuda", 0)
P = lambda t: t.data_ptr()
ORDER = os.environ.get("ORDER", "handle_first")
if ORDER != "handle_first":
    d = [torch.from_numpy(v).to(dev) for v in (th, rp, rd, xp, xd)]
    d_dx = torch.empty(n, dtype=torch.float64, device=dev); d_dy = torch.empty(m, dtype=torch.float64, device=dev)
for N in [int(v) for v in os.environ.get("NLIST", "1,2,4,8").split(",")]:
    kkt = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=row_block, rank=0, nranks=N, streams=int(os.environ.get('NG', '0'))))
    if ORDER == "handle_first":
        d = [torch.from_numpy(v).to(dev) for v in (th, rp, rd, xp, xd)]
        d_dx = torch.empty(n, dtype=torch.float64, device=dev); d_dy = torch.empty(m, dtype=torch.float64, device=dev)
    def step():
        kkt.update_local(P(d[0]), P(d[1]), P(d[2])); kkt.update_finish()
        for _ in range(4):
            kkt.solve_local(P(d[3]), P(d[4])); kkt.solve_finish(P(d_dx), P(d_dy), P(d[4]))
        kkt.sync()
    def fused():
        kkt.update_local(P(d[0]), P(d[1]), P(d[2])); kkt.update_finish()    # rank-local work only: no all-reduce
        for _ in range(4):
            kkt.solve_local(P(d[3]), P(d[4])); kkt.solve_finish(P(d_dx), P(d_dy), P(d[4]))
        kkt.sync()
    for fn in ((fused, step) if N == 1 else (step,)):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"  {fn.__name__}: {ms:.1f} ms/step", flush=True)
    kkt.set_profile(True); step(); kt = kkt.kernel_times(); kkt.set_profile(False)
    rp_, cnt_p = kkt.root_panel(); rr_, cnt_r = kkt.root_rhs()
    print(f"nranks={N}: root panel {cnt_p} doubles, root rhs {cnt_r} doubles; rank 0 owns {kkt.stats()['n_local_blocks']} blocks, {ms:.1f} ms/step local work; "
          f"serialised per class: " + ", ".join(f"{k} {v['ms']:.1f}" for k, v in kt.items()), flush=True)
    kkt.close()
