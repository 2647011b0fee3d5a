"""N > 1 path of the block-angular sharding, on CPU: world_size-2 (and 3) `gloo` processes.

Each rank (tests/dist_worker.py) runs the library's host analyse with its (rank, nranks), then
executes ITS OWN exported schedule in numpy (tests/emulate.py -- the same task lists the HIP
kernels replay) up to the all-reduce marker, all-reduces the root panel / root right-hand side
with torch.distributed, and finishes.  Checked: every rank ends with the oracle's solution of
the whole LP; ownership is a partition; the split-phase sequence is the one bench.py uses."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_newton_step_matches_oracle(world, tmp_path):
    from dist_worker import PROBLEM
    from helpers import block_angular, ipm_like_data
    from oracle_binding import OracleK1
    seed = 11
    port = str(_free_port())
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), port,
                               str(seed), outs[r]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("distributed worker timed out")
        logs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    A, row_block = block_angular(seed=seed, **PROBLEM)
    m, n = A.shape
    th, rp, rd, xp, xd = ipm_like_data(m, n, seed)
    orc = OracleK1(A)
    orc.update(th, rp, rd)
    dxo, dyo = orc.solve(xp, xd)
    nloc = []
    for r in range(world):
        z = np.load(outs[r])
        assert np.abs(z["dx"] - dxo).max() <= 1e-9 * max(1, np.abs(dxo).max())
        assert np.abs(z["dy"] - dyo).max() <= 1e-9 * max(1, np.abs(dyo).max())
        assert (z["own"] == 1).sum() == n + (row_block >= 0).sum()      # ownership is a partition
        assert int(z["rlen"]) == PROBLEM["m0"] ** 2
        assert np.abs(z["dy_link"] - dyo[row_block < 0]).max() <= 1e-9 * max(1, np.abs(dyo).max())  # replicated
        nloc.append(int(z["nloc"]))
        # one refinement step across the shards (residuals formed rank by rank, partial sums on the linking rows completed by the
        # reduction inside the solve): still the oracle's solution, and the residuals of the augmented system do not grow
        from helpers import kkt_residuals
        assert np.abs(z["dx_refined"] - dxo).max() <= 1e-9 * max(1, np.abs(dxo).max())
        assert np.abs(z["dy_refined"] - dyo).max() <= 1e-9 * max(1, np.abs(dyo).max())
        r0 = max(kkt_residuals(A, th, rp, rd, xp, xd, z["dx"], z["dy"]))
        r1 = max(kkt_residuals(A, th, rp, rd, xp, xd, z["dx_refined"], z["dy_refined"]))
        scale = max(np.abs(xp).max(), np.abs(xd).max())
        assert r1 <= max(2.0 * r0, 1e-13 * scale), (r0, r1)
    assert sum(nloc) == PROBLEM["nblocks"] and all(v >= 1 for v in nloc)


def test_eight_ranks_on_the_64_block_partition(tmp_path):
    """The shape an 8-GPU launch of the bench shards (BASELINE configs[3]: 64 diagonal blocks + linking rows; hook
    /root/reference/src/parameters.jl:11, src/IPM/ipmdata.jl:166), at reduced block size on 8 `gloo` processes: the ownership is a partition, every
    rank owns 8 of the 64 blocks with factor flops within 10 % of the mean (the partition balances sum f^2-type flops over contiguous block ranges),
    and every rank ends with the oracle's solution (update: local subtrees + all-reduce of the root panel + replicated root; solve: all-reduce of
    the root right-hand side; one refinement step)."""
    from dist_worker import PROBLEM64
    from helpers import block_angular, ipm_like_data
    from oracle_binding import OracleK1
    world, seed = 8, 23
    port = str(_free_port())
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", TLPK_HOST_THREADS="1")
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), port, str(seed), outs[r], "K1", "p64"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("distributed worker timed out")
        logs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    A, row_block = block_angular(seed=seed, **PROBLEM64)
    m, n = A.shape
    th, rp, rd, xp, xd = ipm_like_data(m, n, seed)
    orc = OracleK1(A)
    orc.update(th, rp, rd)
    dxo, dyo = orc.solve(xp, xd)
    nloc, lfl = [], []
    for r in range(world):
        z = np.load(outs[r])
        assert np.abs(z["dx"] - dxo).max() <= 1e-9 * max(1, np.abs(dxo).max())
        assert np.abs(z["dy"] - dyo).max() <= 1e-9 * max(1, np.abs(dyo).max())
        assert np.abs(z["dx_refined"] - dxo).max() <= 1e-9 * max(1, np.abs(dxo).max())
        assert (z["own"] == 1).sum() == n + (row_block >= 0).sum()      # ownership is a partition
        assert int(z["rlen"]) == PROBLEM64["m0"] ** 2
        nloc.append(int(z["nloc"])); lfl.append(float(z["lflops"]))
    assert sum(nloc) == 64 and all(6 <= v <= 10 for v in nloc), nloc
    assert max(lfl) <= 1.10 * np.mean(lfl) and min(lfl) >= 0.90 * np.mean(lfl), lfl


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_augmented_system_matches_k2_oracle(world, tmp_path):
    """K2 (the reference's default linear system, KKT.jl:134-141) on sharded handles: every rank owns the variable and constraint
    nodes of its diagonal blocks, the root front (linking constraint nodes + the variable nodes of columns that touch linking rows
    only, pivots of both signs) is replicated, its assembled entries and its right-hand side come from rank 0, the ranks'
    contributions arrive through the two all-reduces.  Every rank must end with the K2 oracle's solution."""
    from dist_worker import PROBLEM
    from helpers import block_angular, ipm_like_data
    from oracle_binding import OracleK2
    seed = 17
    port = str(_free_port())
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), port, str(seed), outs[r], "K2"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("distributed worker timed out")
        logs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    A, row_block = block_angular(seed=seed, **PROBLEM)
    m, n = A.shape
    th, rp, rd, xp, xd = ipm_like_data(m, n, seed)
    orc = OracleK2(A)
    orc.update(th, rp, rd)
    dxo, dyo = orc.solve(xp, xd)
    for r in range(world):
        z = np.load(outs[r])
        # dx: every column's variable node is owned by one rank, or sits in the replicated root (then every rank returns it)
        own = z["own"]
        nrep_x = np.where(own[:n] == 0, world, 1)              # a root variable node is reported by every rank
        assert np.abs(z["dx"] / nrep_x - dxo).max() <= 1e-9 * max(1, np.abs(dxo).max())
        assert np.abs(z["dy"] - dyo).max() <= 1e-9 * max(1, np.abs(dyo).max())
        assert np.abs(z["dy_link"] - dyo[row_block < 0]).max() <= 1e-9 * max(1, np.abs(dyo).max())
