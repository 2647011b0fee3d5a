"""Where do the ~0.2 ms of one 256-wide diagonal block (k_potrf_wide) go?  Runs a build of the library with -DPOTRF_TRACE
(tools/_trace/libtlpk_trace.so: the kernel stores 100 MHz time stamps of its phases in the never-read block above the second
diagonal block) on a single dense front and prints the phase durations.
    cd tulip.jl_amd/csrc && for f in *.hip *.cpp; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPOTRF_TRACE -c $f -o /tmp/tr_${f%.*}.o; done
    hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o tools/_trace/libtlpk_trace.so /tmp/tr_*.o"""
import os, sys
import numpy as np
import scipy.sparse as sp
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import tulip_jl_amd._lib as L   # noqa: E402
L.LIB_PATH = os.path.join(ROOT, "tools", "_trace", "libtlpk_trace.so")
import tulip_jl_amd as tk   # noqa: E402

rng = np.random.default_rng(1)
m, n = 600, 1500
A = sp.csc_matrix(rng.standard_normal((m, n)))
kkt = tk.setup(A, tk.K1(), tk.Backend(device=0))
f = kkt.symbolic("front_f"); ns = kkt.symbolic("front_ns"); lda = kkt.symbolic("front_lda"); loff = kkt.symbolic("front_loff")
s = int(np.argmax(ns))
print("fronts:", len(f), "largest: f=%d ns=%d lda=%d" % (f[s], ns[s], lda[s]))
th = 10.0 ** rng.uniform(-2, 2, n)
names = ["potrf 0", "trsm(0) rows 1", "trsm(0) rows 2", "trsm(0) rows 3", "potrf 1 (K=64)", "trsm(1) rows 2", "trsm(1) rows 3", "potrf 2 (K=128)",
         "trsm(2) rows 3", "potrf 3 (K=192)"]
acc = np.zeros(len(names)); reps = 0
for it in range(5):
    tk.update(kkt, th, np.full(n, 1e-6), np.full(m, 1e-6))
    P = kkt.factor_panels()
    for k0 in range(0, int(ns[s]) - 255, 256):
        base = int(loff[s]) + (k0 + 64) * int(lda[s]) + k0
        t = P[base: base + len(names) + 1]
        d = np.diff(t) * 10.0 / 1e3          # 100 MHz ticks -> us
        if it > 0:
            acc += d; reps += 1
        if it == 4:
            print("block column at %d: total %.1f us: " % (k0, d.sum()) + ", ".join("%s %.1f" % (nm, v) for nm, v in zip(names, d)))
print("mean over %d: total %.1f us" % (reps, acc.sum() / reps))
for nm, v in zip(names, acc / reps):
    print("   %-18s %6.1f us" % (nm, v))
