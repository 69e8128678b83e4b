"""N > 1 path on CPU: world_size 2, gloo, oracle standing in for the kernels (host logic only)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from helpers import OracleBackend
    from vivit_amd import kernels
    from vivit_amd.distributed import column_slices, sharded_gram

    kernels.set_backend_for_testing(OracleBackend())
    dims, batch = (9, 8, 4), 5
    facs = bench.mlp_sqrt_ggn_factors(dims, batch, torch.device("cpu"), shard=(rank, world), seed=1)
    G = sharded_gram([f.view(4, batch, -1) for f in facs])
    full = bench.mlp_sqrt_ggn_factors(dims, batch, torch.device("cpu"), seed=1)
    ref = sum(f @ f.T for f in full)
    ok = torch.allclose(G, ref, rtol=1e-5, atol=1e-6)
    # replicated eigensolve gives identical results on all ranks
    w, _ = kernels.symeig(G, eigenvectors=False)
    gathered = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    sl = column_slices(10, 3)
    ok_slices = sl == [(0, 3), (3, 6), (6, 10)]
    # eigensolver with the back-transformation sharded by eigenvector rows + all-gather
    from vivit_amd.distributed import row_slices, symeig as dist_symeig

    w2, Z2 = dist_symeig(G)
    w_ref, Z_ref = kernels.symeig(G, eigenvectors=True)
    ok_eig = torch.equal(w2, w_ref) and torch.equal(Z2, Z_ref) and Z2.shape == Z_ref.shape
    # n not divisible by the world size: padded last slice
    g = torch.Generator().manual_seed(3)
    M = torch.randn(7, 7, generator=g)
    S7 = (M + M.T) / 2
    w7, Z7 = dist_symeig(S7)
    w7r, Z7r = kernels.symeig(S7, eigenvectors=True)
    ok_eig = ok_eig and torch.equal(w7, w7r) and torch.equal(Z7, Z7r) and Z7.shape == (7, 7)
    ok_rows = row_slices(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)] and row_slices(4, 8)[-1] == (4, 4)
    ret[rank] = bool(ok and same and ok_slices and ok_eig and ok_rows)
    dist.destroy_process_group()


def test_sharded_gram_world2():
    world = 2
    port = 29000 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r, False) for r in range(world)), dict(ret)
