#!/usr/bin/env python3
"""Per-launch efficiency of k_update (GPU): duration of every update launch of one KKT.update! in the serialised profile mode
(TLPK_PROF_DUMP), set against the K slabs its tiles execute.  Answers: which launches are short of the matrix-core peak, and is
it the tail (tiles not a multiple of the 512 resident workgroups), the tile mix (few long tiles) or the kernel itself.
    python tools/update_launch_eff.py c4|headline"""
import os, sys, tempfile, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tulip_jl_amd as tk
from workloads import block_angular_lp, kernel_inputs

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
A, rb = block_angular_lp() if which == "c4" else block_angular_lp(100, 20000, 10000, 1000, 4, 0.5, ineq=True)
m, n = A.shape
dump = tempfile.mktemp(prefix="tlpk_prof_")
os.environ["TLPK_PROF_DUMP"] = dump
kkt = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb))
th, rp, rd, xp, xd = kernel_inputs(m, n, 7, "mid")
dev = torch.device("cuda", 0)
d = [torch.from_numpy(v).to(dev) for v in (th, rp, rd)]
P = lambda t: t.data_ptr()
for _ in range(2): kkt.update_device(P(d[0]), P(d[1]), P(d[2])); kkt.sync()
kkt.set_profile(True)
open(dump, "w").close()
kkt.update_device(P(d[0]), P(d[1]), P(d[2])); kkt.sync(); kkt.kernel_times()
kkt.set_profile(False)
rows = [l.split() for l in open(dump) if not l.startswith("#")]
ut = kkt.symbolic("update_tasks").reshape(-1, 10)
SLOTS, PEAK = 512, 78.6e12
print(f"{which}: update launches of one KKT.update! (serialised; slab = 16 K columns of a 128 x 128 tile = 524 288 flops)")
print("%4s %7s %7s %8s %8s %8s %9s %8s %8s %8s" % ("#", "tiles", "rounds", "slabs/t", "max", "ms", "TFLOP/s", "frac", "tail ms", "pack ms"))
tot_ms = tot_fl = tot_tail = tot_pack = 0.0; k = 0
for cls, kind, first, count, ms in rows:
    if int(kind) != 3: continue
    first, count, ms = int(first), int(count), float(ms)
    t = ut[first:first + count]
    slabs = np.where(t[:, 8] > 0, t[:, 9] + (t[:, 2] % 16 > 0), (t[:, 2] + 15) // 16).astype(float)
    fl = slabs.sum() * 2 * 128 * 128 * 16
    # time a perfectly packed launch of these tiles would take at THIS launch's achieved per-slot rate: the sum of slabs / 512 slots against the
    # longest slot of a greedy list schedule (what the hardware dispatcher does)

    h = [0.0] * min(SLOTS, count); heapq.heapify(h)
    for x in slabs + 3.0: heapq.heappush(h, heapq.heappop(h) + x)
    mk = max(h); ideal = (slabs + 3.0).sum() / SLOTS
    tail = ms * (1 - ideal / mk)
    tot_ms += ms; tot_fl += fl; tot_tail += tail
    if ms > 0.25:
        print("%4d %7d %7.2f %8.1f %8.0f %8.3f %9.1f %8.3f %8.3f" % (k, count, count / SLOTS, slabs.mean(), slabs.max(), ms, fl / ms / 1e9, fl / ms / 1e-3 / PEAK, tail))
    k += 1
print(f"total: {k} launches, {tot_ms:.2f} ms, executed {tot_fl / 1e12:.3f} TFLOP = {tot_fl / tot_ms / 1e9:.1f} TFLOP/s ({tot_fl / tot_ms / 1e-3 / PEAK:.3f} of peak); "
      f"list-schedule tails (model) {tot_tail:.2f} ms")
os.unlink(dump)
