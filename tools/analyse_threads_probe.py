#!/usr/bin/env python3
"""Host analyse time (tlpk_stats.ms_analyse) against the number of host threads, on the box it runs on (no GPU work).
    python tools/analyse_threads_probe.py c4|headline"""
import os, subprocess, sys
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
if len(sys.argv) > 2:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import tulip_jl_amd as tk
    from workloads import block_angular_lp
    A, rb = block_angular_lp() if which == "c4" else block_angular_lp(100, 20000, 10000, 1000, 4, 0.5, ineq=True)
    best = min(tk.setup(A, tk.K1(), tk.Backend(device=-1, row_block=rb)).stats()["ms_analyse"] for _ in range(2))
    print("%s  TLPK_HOST_THREADS=%s  ms_analyse %.0f" % (which, os.environ.get("TLPK_HOST_THREADS", "(default)"), best))
else:
    try: print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
    except OSError: pass
    for t in (None, "8", "16", "24", "32", "64"):
        env = dict(os.environ)
        if t: env["TLPK_HOST_THREADS"] = t
        else: env.pop("TLPK_HOST_THREADS", None)
        subprocess.run([sys.executable, __file__, which, "child"], env=env)
