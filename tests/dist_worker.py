"""One rank of the CPU (gloo) sharding test -- launched by tests/test_distributed.py as a plain
subprocess:  python dist_worker.py RANK WORLD PORT SEED OUT.npz"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PROBLEM = dict(nblocks=5, mk=40, nk=90, m0=14, nnz_in=3, link_prob=0.6)
# the 64-block partition of the bench workload (BASELINE configs[3]) at reduced block size: what an 8-rank launch shards (argv[7] = "p64")
PROBLEM64 = dict(nblocks=64, mk=30, nk=64, m0=24, nnz_in=3, link_prob=0.5)


def main():
    rank, world, port, seed, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    system = sys.argv[6] if len(sys.argv) > 6 else "K1"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # the container hostname may not resolve
    import numpy as np
    import torch
    import torch.distributed as dist
    import tulip_jl_amd as tk
    from emulate import Emulator
    from helpers import block_angular, ipm_like_data
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        problem = PROBLEM64 if (len(sys.argv) > 7 and sys.argv[7] == "p64") else PROBLEM
        A, row_block = block_angular(seed=seed, **problem)
        m, n = A.shape
        th, rp, rd, xp, xd = ipm_like_data(m, n, seed)
        kkt = tk.setup(A, tk.K2() if system == "K2" else tk.K1(), tk.Backend(device=-1, row_block=row_block, rank=rank, nranks=world))
        em = Emulator(kkt)
        # --- update: local subtrees + partial root panel, all-reduce, root factorisation ---
        em.update(th, rp, rd, stop_at_marker=True)
        panel = torch.from_numpy(em.root_panel())          # shares memory with the emulator's Lval
        dist.all_reduce(panel)
        em.update_finish()
        assert em.fail_col is None
        # --- solve ---
        em.solve_local(xp, xd, A)
        rhs = torch.from_numpy(em.root_rhs())
        dist.all_reduce(rhs)
        dx, dy = em.solve_finish(xd, A)
        extra = {}
        if system == "K1":
            # one iterative-refinement step, split like the solve (tlpk_refine_local -> all-reduce of the root rhs -> tlpk_refine_finish)
            em.refine_local(dx, dy, xp, xd, A, th, rp, rd)
            rhs = torch.from_numpy(em.root_rhs())
            dist.all_reduce(rhs)
            dxr, dyr = em.refine_finish(dx, dy, A)
            tdxr = torch.from_numpy(dxr.copy()); dist.all_reduce(tdxr)
            dyr_sh = dyr.copy(); dyr_sh[row_block < 0] /= world
            tdyr = torch.from_numpy(dyr_sh); dist.all_reduce(tdyr)
            extra = dict(dx_refined=tdxr.numpy(), dy_refined=tdyr.numpy())
        # assemble the global solution: block rows/cols from their owner, linking rows replicated
        tdx = torch.from_numpy(dx.copy()); dist.all_reduce(tdx)
        link = row_block < 0
        dy_sh = dy.copy(); dy_sh[link] /= world
        tdy = torch.from_numpy(dy_sh); dist.all_reduce(tdy)
        if system == "K2":       # node ownership: variable and constraint nodes outside the replicated root
            own = torch.from_numpy((kkt.symbolic("row_local") == 1).astype(np.int64))
        else:
            own = torch.from_numpy(np.concatenate([(kkt.symbolic("col_local") != 0).astype(np.int64),
                                                   (kkt.symbolic("row_local") == 1).astype(np.int64)]))
        dist.all_reduce(own)
        st = kkt.stats()
        # factor flops of the fronts this rank owns (root excluded): what the partition balances
        fl, fns, floc = kkt.symbolic("front_f").astype(float), kkt.symbolic("front_ns").astype(float), kkt.symbolic("front_local")
        sel = (floc != 0); root = int(kkt.symbolic("root_front")[0])
        if root >= 0:
            sel[root] = False
        lflops = float((fns[sel] ** 3 / 3 + (fl[sel] - fns[sel]) * fns[sel] ** 2 + (fl[sel] - fns[sel]) ** 2 * fns[sel]).sum())
        np.savez(out, dx=tdx.numpy(), dy=tdy.numpy(), own=own.numpy(), nloc=st["n_local_blocks"],
                 rlen=st["root_panel_len"], dy_link=dy[link], lflops=lflops, **extra)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
