import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'./tests')
import tulip_jl_amd as tk
from workloads import block_angular_lp
A,rb = block_angular_lp(64,5000,10000,1000,4,0.5)
t0=time.perf_counter()
k = tk.setup(A, tk.K1(), tk.Backend(device=-1, row_block=rb))
print("setup wall", time.perf_counter()-t0, "ms_analyse", k.stats()["ms_analyse"])
