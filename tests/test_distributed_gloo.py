"""N > 1 path on CPU: world_size 2 (and 3), gloo, oracle standing in for the kernels (host logic only)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import OracleBackend
    from vivit_amd import kernels

    kernels.set_backend_for_testing(OracleBackend())


def _worker_param_sharded(rank, world, port, ret):
    _setup(rank, world, port)
    import bench
    from vivit_amd import kernels
    from vivit_amd.distributed import column_slices, sharded_gram

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


def _worker_batch_sharded(rank, world, port, ret):
    """Data-parallel layout: every rank holds the factors of ITS samples only; the assembled Gram
    matrix, V^T g and back-projections must equal the single-process ones in the reference's class-major layout."""
    _setup(rank, world, port)
    from oracle import vivit_oracle as oracle
    from vivit_amd import distributed as vd

    C, Ng = 3, 4
    N = Ng * world
    g = torch.Generator().manual_seed(11)
    # two materialised parameters (one big enough to be column-sliced, one small -> owned by one rank), one Linear
    vd.SMALL_PARAM_COLUMNS = 8
    V1 = torch.randn(C, N, 6, 7, generator=g)       # 42 columns: sliced over the ranks
    V2 = torch.randn(C, N, 5, generator=g)          # 5 columns: owned whole by rank 1 % world
    g1 = torch.randn(N, 6, 7, generator=g)
    g2 = torch.randn(N, 5, generator=g)
    s = torch.randn(C, N, 5, generator=g)
    z = torch.randn(N, 9, generator=g)
    delta = torch.randn(N, 5, generator=g)
    lo, hi = rank * Ng, (rank + 1) * Ng
    ok = {}

    # (1) materialised factors: all-to-all -> parameter shards -> partial SYRK -> all-reduce
    G = vd.batch_sharded_gram([V1[:, lo:hi], V2[:, lo:hi]])
    ref = oracle.compute_gram_mat([V1, V2], start_dim=2, flatten=True)
    ok["alltoall_gram"] = torch.allclose(G, ref, rtol=1e-5, atol=1e-5)

    # (2) everything at once: V^T V and V^T g of materialised (both kinds), small (rows path) and factorised Linear
    acc = vd.BatchShardedGram(C, Ng, N_grad_local=Ng)
    acc.add_factor(V1[:, lo:hi], g1[lo:hi])
    acc.add_factor_rows(V2[:, lo:hi], g2[lo:hi])
    acc.add_linear(s[:, lo:hi], z[lo:hi], delta[lo:hi])
    Gall = acc.finalize()
    VtG = acc.finalize_vtg()
    Vw = torch.einsum("cno,ni->cnoi", s, z)
    gw = torch.einsum("no,ni->noi", delta, z)
    ref_all = oracle.compute_gram_mat([V1, V2, Vw], start_dim=2, flatten=False)
    ref_vtg = sum(oracle.partial_contract(V, gg, (2, 1)) for V, gg in [(V1, g1), (V2, g2), (Vw, gw)])
    ok["mixed_gram"] = Gall.shape == (C, N, C, N) and torch.allclose(Gall, ref_all, rtol=1e-5, atol=1e-4)
    ok["mixed_vtg"] = VtG.shape == (C, N, N) and torch.allclose(VtG, ref_vtg, rtol=1e-5, atol=1e-4)

    # (3) factorised only (BASELINE config 5 layout, C = 1 and C > 1): block rows + all-gather, class-major store
    for CC in (1, C):
        acc = vd.BatchShardedGram(CC, Ng)
        acc.add_linear(s[:CC, lo:hi], z[lo:hi])
        Gl = acc.finalize()
        ok[f"linear_gram_C{CC}"] = torch.allclose(Gl, oracle.linear_weight_gram(s[:CC], z), rtol=1e-5, atol=1e-5)

    # (4) back-projection: own rows of V, one all-reduce of K * P floats
    coef = torch.randn(2, C, N, generator=g)
    acc = vd.BatchShardedGram(C, Ng)
    step = vd.backproject_sum(coef, V1[:, lo:hi], acc)
    ok["backproject"] = torch.allclose(step, oracle.Vmp(V1, coef, 2), rtol=1e-5, atol=1e-5)

    # (5) every rank holds the same result
    flat = Gall.reshape(-1)
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok["replicated"] = all(torch.equal(gathered[0], t) for t in gathered)

    # (6) unequal shards are rejected
    try:
        vd.check_equal_shards(Ng + rank)
        ok["unequal_raises"] = False
    except ValueError:
        ok["unequal_raises"] = True
    ok["equal_ok"] = vd.check_equal_shards(Ng) == N
    ret[rank] = {k: bool(v) for k, v in ok.items()}
    dist.destroy_process_group()


def _run(worker, world):
    port = 29000 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, port, ret), nprocs=world, join=True)
    return dict(ret)


def test_sharded_gram_world2():
    ret = _run(_worker_param_sharded, 2)
    assert all(ret.get(r, False) for r in range(2)), ret


@pytest.mark.parametrize("world", [2, 3])
def test_batch_sharded_gram(world):
    ret = _run(_worker_batch_sharded, world)
    for r in range(world):
        assert r in ret and all(ret[r].values()), ret
