"""Timeline analysis of a rocprofv3 kernel trace of the default (concurrent) schedule: for one Newton step, how long is the device busy
with each kernel class ALONE, how much of the non-update work is hidden behind k_update, where are the gaps.
    python tools/timeline_overlap.py gpurun_out/r03_trace_c4_kernels.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
norm = lambda n: n.replace("void ", "").replace("tlpk::", "")
ev = [(norm(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")) for r in rows]
ev.sort(key=lambda e: e[1])
starts = [i for i, e in enumerate(ev) if e[0].startswith("k_compute_d")]
if len(starts) < 2:
    sys.exit("need two steps in the trace")
# STEP=<k>: the k-th step of the trace (default: the last complete one); the list of steps goes to stderr
for i in range(len(starts) - 1):
    print("step %d: %d kernels, %.2f ms" % (i, starts[i + 1] - starts[i], (ev[starts[i + 1]][1] - ev[starts[i]][1]) / 1e6), file=sys.stderr)
import os
k = int(os.environ.get("STEP", len(starts) - 2))
a, b = starts[k], starts[k + 1]
step = ev[a:b]
t0 = step[0][1]; t1 = max(e[2] for e in step)
def cls(n):
    for k in ("k_update_reduce", "k_update", "k_extend_add", "k_trsm", "k_potrf", "k_fwd", "k_bwd", "k_zero", "k_assemble"):
        if n.startswith(k): return k
    return "other"
# sweep line
pts = []
for n, s, e, q in step:
    pts.append((s, 1, cls(n))); pts.append((e, -1, cls(n)))
pts.sort()
active = collections.Counter(); last = t0
alone = collections.Counter(); combo = collections.Counter(); idle = 0
for t, d, c in pts:
    dt = t - last
    if dt > 0:
        live = tuple(sorted(k for k, v in active.items() if v > 0))
        if not live: idle += dt
        elif len(live) == 1: alone[live[0]] += dt
        combo[live] += dt
    active[c] += d; last = t
tot = collections.Counter()
for n, s, e, q in step: tot[cls(n)] += e - s
ms = lambda x: x / 1e6
print("one Newton step (1 update + 4 right-hand sides): %d kernels, span %.2f ms, device idle (no kernel running) %.2f ms" % (len(step), ms(t1 - t0), ms(idle)))
print("%-16s %10s %12s" % ("class", "sum of durations", "running ALONE"))
for k, v in tot.most_common(): print("%-16s %10.2f ms %10.2f ms" % (k, ms(v), ms(alone[k])))
print("time with k_update running: %.2f ms" % ms(sum(v for c, v in combo.items() if "k_update" in c)))
print("largest combinations:")
for c, v in combo.most_common(12): print("   %6.2f ms  %s" % (ms(v), " + ".join(c) if c else "(idle)"))
queues = collections.Counter(e[3] for e in step)
print("kernels per queue:", dict(queues))
