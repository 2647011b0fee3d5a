#!/usr/bin/env python3
"""Kernel sequence of the solves of one Newton step, from a rocprofv3 kernel trace of the concurrent schedule (tools/gpu_trace.sh):
start, in-stream gap, duration, queue and workgroups of every kernel, and the time per phase of the last single solve of the step.
    STEP=3 python tools/solve_timeline.py gpurun_out/r04_trace_c4_kernels.csv"""
import csv, os, sys
rows = list(csv.DictReader(open(sys.argv[1])))
norm = lambda n: n.replace("void ", "").replace("tlpk::", "").split("(")[0]
ev = [(norm(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"],
       int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows]
ev.sort(key=lambda e: e[1])
starts = [i for i, e in enumerate(ev) if e[0].startswith("k_compute_d")]
k = int(os.environ.get("STEP", 3))
step = ev[starts[k]:starts[k + 1]]
last_f = max(i for i, e in enumerate(step) if e[0].startswith(("k_update", "k_trsm", "k_potrf", "k_extend")))
sol = step[last_f + 1:]
t0 = sol[0][1]
print("solves of step %d: %d kernels, %.2f ms (1 pair of right-hand sides + 2 single solves)" % (k, len(sol), (max(e[2] for e in sol) - t0) / 1e6))
prev_end = t0
for n, s, e, q, g in sol:
    print("%8.1f us  %+7.1f us after the previous end  %7.1f us  queue %s  %6d workgroups  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, g, n))
    prev_end = max(prev_end, e)
# phases of the LAST single solve: from its fill kernel to its k_dx
idx = [i for i, e in enumerate(sol) if e[0] == "k_dx"]
a = max(i for i, e in enumerate(sol[:idx[-1]]) if e[0].startswith("__amd_rocclr_fillBuffer"))
one = sol[a:idx[-1] + 1]
def span(pred):
    sel = [e for e in one if pred(e)]
    return (min(e[1] for e in sel), max(e[2] for e in sel)) if sel else (0, 0)
big_f = [e for e in one if e[0].startswith("k_fwd_sweep") and e[4] > 1500]
big_b = [e for e in one if e[0].startswith("k_bwd_sweep") and e[4] > 1500]
T0, T1 = one[0][1], one[-1][2]
f0, f1 = min(e[1] for e in big_f), max(e[2] for e in big_f)
b0, b1 = min(e[1] for e in big_b), max(e[2] for e in big_b)
first_fwd = min(e[1] for e in one if e[0].startswith("k_fwd"))
last_bwd = max(e[2] for e in one if e[0].startswith("k_bwd"))
print("\nlast single solve: %.1f us" % ((T1 - T0) / 1e3))
print("  right-hand side (fill, k_rhs_scale, k_rhs, k_single_solve)          %7.1f us" % ((first_fwd - T0) / 1e3))
print("  forward, levels below the diagonal blocks' top fronts (gather/small/sweep launches) %7.1f us" % ((f0 - first_fwd) / 1e3))
print("  forward sweeps of the block level (two stream groups side by side)   %7.1f us" % ((f1 - f0) / 1e3))
print("  root front: gather, forward sweep, backward sweep                    %7.1f us" % ((b0 - f1) / 1e3))
print("  backward sweeps of the block level                                   %7.1f us" % ((b1 - b0) / 1e3))
print("  backward, levels below (overlapping the tail of the block level)     %7.1f us" % ((last_bwd - b1) / 1e3))
print("  k_unpermute, k_dx                                                    %7.1f us" % ((T1 - last_bwd) / 1e3))
