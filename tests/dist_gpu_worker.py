"""One rank of the GPU sharding test: several ranks share ONE GPU (gpurun boxes have a single
MI355X), run the real HIP split-phase path, and all-reduce the root panel / rhs through gloo on
host copies (RCCL refuses two ranks on one device; bench.py uses backend nccl on real multi-GPU
nodes).  python dist_gpu_worker.py RANK WORLD PORT SEED OUT.npz"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PROBLEM = dict(nblocks=6, mk=300, nk=600, m0=70, nnz_in=3, link_prob=0.5)


def main():
    rank, world, port, seed, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import numpy as np
    import torch
    import torch.distributed as dist
    import tulip_jl_amd as tk
    from helpers import block_angular, ipm_like_data
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        A, row_block = block_angular(seed=seed, **PROBLEM)
        m, n = A.shape
        th, rp, rd, xp, xd = ipm_like_data(m, n, seed)
        kkt = tk.setup(A, tk.K1(), tk.Backend(device=0, row_block=row_block, rank=rank, nranks=world))
        d = [torch.from_numpy(v).to(dev) for v in (th, rp, rd, xp, xd)]
        d_dx = torch.empty(n, dtype=torch.float64, device=dev)
        d_dy = torch.empty(m, dtype=torch.float64, device=dev)
        P = lambda t: t.data_ptr()   # noqa: E731

        def allreduce_device(which, count):
            # torch-owned staging tensor <- library (D2D on the library stream), reduce over gloo on
            # a host copy (several ranks share this GPU, RCCL needs one device per rank), copy back
            if not count:
                return
            buf = torch.empty(count, dtype=torch.float64, device=dev)
            kkt.root_copy(which, "out", P(buf))
            kkt.sync()
            h = buf.cpu()
            dist.all_reduce(h)
            buf.copy_(h)
            torch.cuda.synchronize()
            kkt.root_copy(which, "in", P(buf))
            kkt.sync()

        # the composed entry points would skip the all-reduce of the root panel / rhs on a sharded
        # handle and return a silently wrong factor: they must refuse (TLPK_BADARG -> DimensionMismatch)
        for call in (lambda: tk.update(kkt, th, rp, rd), lambda: kkt.update_device(P(d[0]), P(d[1]), P(d[2]))):
            try:
                call()
            except tk.DimensionMismatch as e:
                assert "split-phase" in str(e)
            else:
                raise AssertionError("composed update on a sharded handle did not fail")
        try:
            kkt.solve_finish(P(d_dx), P(d_dy), P(d[4]))          # before any update / solve_local
        except (RuntimeError, tk.DimensionMismatch):
            pass
        else:
            raise AssertionError("solve_finish without solve_local did not fail")
        kkt.update_local(P(d[0]), P(d[1]), P(d[2]))
        kkt.sync()
        allreduce_device("panel", kkt.root_panel()[1])
        kkt.update_finish()
        try:
            kkt.solve_finish(P(d_dx), P(d_dy), P(d[4]))          # factored, but no solve_local yet
        except tk.DimensionMismatch as e:
            assert "tlpk_solve_local" in str(e)
        else:
            raise AssertionError("solve_finish without solve_local did not fail")
        kkt.solve_local(P(d[3]), P(d[4]))
        kkt.sync()
        allreduce_device("rhs", kkt.root_rhs()[1])
        kkt.solve_finish(P(d_dx), P(d_dy), P(d[4]))
        kkt.sync()
        dx, dy = d_dx.cpu(), d_dy.cpu()
        link = torch.from_numpy(row_block < 0)
        dy_link = dy[link].numpy().copy()
        dy = torch.where(link, dy / world, dy)
        dist.all_reduce(dx); dist.all_reduce(dy)
        # a PAIR of solves in one pass over this rank's factor (tlpk_solve2_local -> all-reduce of BOTH root right-hand sides ->
        # tlpk_solve2_finish): bit-identical to two split solves with the same reduction
        xp2 = torch.from_numpy(xp[::-1].copy()).to(dev); xd2 = torch.from_numpy(xd[::-1].copy()).to(dev)
        s_dx = torch.empty_like(d_dx); s_dy = torch.empty_like(d_dy)
        kkt.solve_local(P(xp2), P(xd2)); kkt.sync()
        allreduce_device("rhs", kkt.root_rhs()[1])
        kkt.solve_finish(P(s_dx), P(s_dy), P(xd2)); kkt.sync()
        p_dx = [torch.empty_like(d_dx) for _ in range(2)]; p_dy = [torch.empty_like(d_dy) for _ in range(2)]
        kkt.solve2_local(P(d[3]), P(d[4]), P(xp2), P(xd2)); kkt.sync()
        allreduce_device("rhs", kkt.root_rhs()[1])
        allreduce_device("rhs2", kkt.root_rhs2()[1])
        kkt.solve2_finish(P(p_dx[0]), P(p_dy[0]), P(d[4]), P(p_dx[1]), P(p_dy[1]), P(xd2)); kkt.sync()
        assert torch.equal(p_dx[0], d_dx) and torch.equal(p_dy[0], d_dy), "pair, first system"
        assert torch.equal(p_dx[1], s_dx) and torch.equal(p_dy[1], s_dy), "pair, second system"
        try:
            kkt.solve2_finish(P(p_dx[0]), P(p_dy[0]), P(d[4]), P(p_dx[1]), P(p_dy[1]), P(xd2))
        except tk.DimensionMismatch as e:
            assert "tlpk_solve2_local" in str(e)
        else:
            raise AssertionError("solve2_finish without solve2_local did not fail")
        # one iterative-refinement step, split like the solve: residuals rank by rank (partial sums on the linking rows), the
        # reduction of the root right-hand side completes them
        kkt.refine_local(P(d_dx), P(d_dy), P(d[3]), P(d[4]))
        kkt.sync()
        allreduce_device("rhs", kkt.root_rhs()[1])
        kkt.refine_finish(P(d_dx), P(d_dy))
        kkt.sync()
        dxr, dyr = d_dx.cpu(), d_dy.cpu()
        dyr_link = dyr[link].numpy().copy()
        dyr = torch.where(link, dyr / world, dyr)
        dist.all_reduce(dxr); dist.all_reduce(dyr)
        np.savez(out, dx=dx.numpy(), dy=dy.numpy(), dy_link=dy_link, nloc=kkt.stats()["n_local_blocks"],
                 dx_refined=dxr.numpy(), dy_refined=dyr.numpy(), dy_link_refined=dyr_link)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
