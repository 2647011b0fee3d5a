import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tulip_jl_amd as tk
from helpers import block_angular, ipm_like_data
A, rb = block_angular(nblocks=8, mk=300, nk=600, m0=60, nnz_in=3, link_prob=0.5, seed=5)
kkt = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb))
f = kkt.symbolic("front_f"); ns = kkt.symbolic("front_ns"); loff = kkt.symbolic("front_loff")
th, rp, rd, xp, xd = ipm_like_data(kkt.m, kkt.n, 3)
try:
    tk.update(kkt, th, rp, rd); print("update ok")
except Exception as e:
    print("update failed:", e)
P = kkt.factor_panels()
np.save(sys.argv[1], P)
if len(sys.argv) > 2:
    Q = np.load(sys.argv[2])
    d = np.nonzero(~((P == Q) | (np.isnan(P) & np.isnan(Q))))[0]
    print("differing entries:", len(d))
    if len(d):
        order = np.argsort(loff)
        fr = order[np.searchsorted(loff[order], d, side="right") - 1]
        for s in np.unique(fr)[:10]:
            ds = d[fr == s] - loff[s]
            print(" front", int(s), "f", int(f[s]), "ns", int(ns[s]), "loff", int(loff[s]), "n", len(ds), "first offsets", ds[:6].tolist(), "vals new/old", P[d[fr == s][:3]].tolist(), Q[d[fr == s][:3]].tolist())
