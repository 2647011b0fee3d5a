import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import tulip_jl_amd as tk
from helpers import block_angular, ipm_like_data
from oracle_binding import OracleK1
A, rb = block_angular(nblocks=7, mk=300, nk=600, m0=70, nnz_in=3, link_prob=0.5, seed=13)
m, n = A.shape
th, rp, rd, xp, xd = ipm_like_data(m, n, 13)
orc = OracleK1(A); orc.update(th, rp, rd); dxo, dyo = orc.solve(xp, xd)
def report(tag, dx, dy):
    ex = np.abs(dx - dxo); ey = np.abs(dy - dyo)
    print(tag, "dx err by block:", [float(ex[k*600:(k+1)*600].max()) for k in range(7)], "dy err by block:", [float(ey[k*300:(k+1)*300].max()) for k in range(7)], "link", float(ey[2100:].max()))
# (a) manual two handles in one process, explicit syncs, host reduce
dev = torch.device("cuda", 0)
W = 2
ks = [tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb, rank=r, nranks=W)) for r in range(W)]
d = [torch.from_numpy(v).to(dev) for v in (th, rp, rd, xp, xd)]
P = lambda t: t.data_ptr()
for k in ks: k.update_local(P(d[0]), P(d[1]), P(d[2])); k.sync()
bufs = []
for k in ks:
    cnt = k.root_panel()[1]; b = torch.empty(cnt, dtype=torch.float64, device=dev); k.root_copy("panel", "out", P(b)); k.sync(); bufs.append(b)
tot = sum(bufs)
for k in ks: k.root_copy("panel", "in", P(tot)); k.sync(); k.update_finish()
for k in ks: k.solve_local(P(d[3]), P(d[4])); k.sync()
bufs = []
for k in ks:
    cnt = k.root_rhs()[1]; b = torch.empty(cnt, dtype=torch.float64, device=dev); k.root_copy("rhs", "out", P(b)); k.sync(); bufs.append(b)
tot = sum(bufs)
outs = []
for k in ks:
    k.root_copy("rhs", "in", P(tot)); k.sync()
    ddx = torch.empty(n, dtype=torch.float64, device=dev); ddy = torch.empty(m, dtype=torch.float64, device=dev)
    k.solve_finish(P(ddx), P(ddy), P(d[4])); k.sync(); outs.append((ddx.cpu().numpy(), ddy.cpu().numpy()))
dx = outs[0][0] + outs[1][0]; dy = outs[0][1] + outs[1][1]; dy[rb < 0] /= 2
report("manual", dx, dy)
# (b) library multi mode
kk = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=rb, ngpus=2, devices=[0, 0]))
tk.update(kk, th, rp, rd)
dx = np.full(n, np.nan); dy = np.full(m, np.nan)
tk.solve(dx, dy, kk, xp, xd)
report("multi ", dx, dy)
tk.solve(dx, dy, kk, xp, xd)
report("multi2", dx, dy)
