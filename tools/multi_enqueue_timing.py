"""Round 5: host-side cost of a Newton step of a `tlpk_create_multi` handle with NSHARDS shards (all on this box's one GPU unless DEVICES lists
ordinals) on the north-star shape (HEADLINE=1, default) or config C4: the host time until every shard's update! is enqueued
(tlpk_stats.ms_enqueue_update), the whole update!, and one solve! -- through the host-pointer calls a Julia process makes.  With all shards on
one GPU the DEVICE times mean little (the shards share the chip); the host times are what an 8-GPU node will see.
    NSHARDS=8 python tools/multi_enqueue_timing.py          TLPK_SHARD_THREADS=0 | TLPK_MULTI_REDUCE=gather for the A/B"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import tulip_jl_amd as tk   # noqa: E402
from workloads import block_angular_lp, kernel_inputs   # noqa: E402

N = int(os.environ.get("NSHARDS", "8"))
devices = [int(d) for d in os.environ.get("DEVICES", ",".join(["0"] * N)).split(",")]
headline = os.environ.get("HEADLINE", "1") == "1"
A, rb = block_angular_lp(100, 20000, 10000, 1000, 4, 0.5, ineq=True) if headline else block_angular_lp()
m, n = A.shape
th, rp, rd, xp, xd = kernel_inputs(m, n, 7, "mid")
t0 = time.perf_counter()
kkt = tk.setup(A, tk.K1(), tk.Backend(device=devices[0], row_block=rb, ngpus=N, devices=devices))
t_setup = time.perf_counter() - t0
dx, dy = np.empty(n), np.empty(m)
enq, upd, sol = [], [], []
for rep in range(4):
    t0 = time.perf_counter(); tk.update(kkt, th, rp, rd); upd.append(1e3 * (time.perf_counter() - t0))
    enq.append(kkt.stats()["ms_enqueue_update"])
    t0 = time.perf_counter(); tk.solve(dx, dy, kkt, xp, xd); sol.append(1e3 * (time.perf_counter() - t0))
r_p = float(np.abs(A @ dx + rd * dy - xp).max()); r_d = float(np.abs(-dx * (th + rp) + A.T @ dy - xd).max())
print(f"{'north-star' if headline else 'C4'} shape, {N} shards on devices {devices}: setup {t_setup:.1f} s; per update! host enqueue "
      f"{min(enq[1:]):.2f} ms of {min(upd[1:]):.1f} ms; solve! {min(sol[1:]):.1f} ms (host vectors, {8 * (2 * n + m) / 1e6:.0f} + {16 * (m + n) / 1e6:.0f} MB over PCIe); "
      f"residuals {r_p:.1e} {r_d:.1e}; TLPK_SHARD_THREADS={os.environ.get('TLPK_SHARD_THREADS', '1')} TLPK_MULTI_REDUCE={os.environ.get('TLPK_MULTI_REDUCE', 'rs')}", flush=True)
kkt.close()
