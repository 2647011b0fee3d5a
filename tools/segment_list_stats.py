import sys, numpy as np, collections
sys.path.insert(0,'/root/repo')
import tulip_jl_amd as tk
from workloads import block_angular_lp
A, rb = block_angular_lp(4, 20000, 10000, 1000, 4, 0.5, ineq=True)
k = tk.setup(A, tk.K1(), tk.Backend(device=-1, row_block=rb))
ut = k.symbolic("update_tasks").reshape(-1,10)
seg = k.symbolic("upd_seg")
fl = k.symbolic("factor_launches").reshape(-1,3)
print("tasks", len(ut), "with seg", (ut[:,8]>0).sum())
def sig(t):
    s=t[8]
    if s==0: return ('full', t[1], t[2])
    n=seg[s-1]; return tuple(seg[s:s+2*n])
tot=0; 
for L in fl:
    if L[0]!=3: continue
    first,cnt=L[1],L[2]
    if cnt<500: continue
    T=ut[first:first+cnt]
    byfront=collections.defaultdict(list)
    for t in T: byfront[t[0]].append(t)
    ng=0; nt=0; sizes=[]
    for f,ts in byfront.items():
        c=collections.Counter(sig(t) for t in ts)
        ng+=len(c); nt+=len(ts); sizes+=list(c.values())
    # locality: consecutive tasks sharing i0 or j0 and equal signature
    same=sum(1 for a,b in zip(T[:-1],T[1:]) if a[0]==b[0] and sig(a)==sig(b))
    print(f"launch first={first} tiles={cnt} fronts={len(byfront)} distinct sigs per front avg={ng/len(byfront):.1f} tiles/front={nt/len(byfront):.1f} median group={np.median(sizes):.0f} max group={max(sizes)} adjacent-equal={same/cnt:.2f}")
