"""The data-parallel path on the HIP kernels: two ranks share the one GPU of the test box (gloo; collectives on device
tensors are staged through the host by vivit_amd.distributed), every rank back-propagates ITS batch shard on cuda:0.
Results must equal the single-process HIP run on the whole batch (same kernels, different assembly order)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _one_card_env():
    """Two ranks share ONE card here.  The one-XCD persistent kernels need their 32 workgroups co-resident on XCD 0, one
    per CU: two processes launching them at the same moment can each hold part of the XCD and wait for the rest until
    the bounded spins give up.  The product runs one process per GPU; this test plumbing takes the launch chains."""
    os.environ["VIVIT_SYTRD_PERSIST"] = "0"
    os.environ["VIVIT_QR_PERSIST"] = "0"


def _worker(rank, world, port, ret):
    _one_card_env()
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vivit_amd
    from helpers import constant_damping, top_k_criterion
    from torch import nn
    from vivit_amd import distributed as vd
    from vivit_amd.backend import backpack, extend

    dev = torch.device("cuda:0")
    ok = {}
    # (1) raw accumulators: all-to-all path, block rows, factorised Linear, V^T g -- against one-process kernels
    C, Ng = 10, 96
    N = Ng * world
    g = torch.Generator().manual_seed(3)
    V1 = torch.randn(C, N, 40, 130, generator=g).to(dev)     # 5200 columns: all-to-all to parameter shards
    V2 = torch.randn(C, N, 40, generator=g).to(dev)          # narrow: block rows
    s = torch.randn(C, N, 24, generator=g).to(dev)
    z = torch.randn(N, 50, generator=g).to(dev)
    g1 = torch.randn(N, 40, 130, generator=g).to(dev)
    lo, hi = rank * Ng, (rank + 1) * Ng
    acc = vd.BatchShardedGram(C, Ng, N_grad_local=Ng)
    acc.add_factor(V1[:, lo:hi].contiguous(), g1[lo:hi].contiguous())
    acc.add_factor_rows(V2[:, lo:hi].contiguous())
    acc.add_linear(s[:, lo:hi].contiguous(), z[lo:hi].contiguous())
    G = acc.finalize().reshape(C * N, C * N)
    VtG = acc.finalize_vtg().reshape(C * N, N)
    Vw = torch.einsum("cno,ni->cnoi", s, z)
    ref = sum((F.reshape(C * N, -1).double() @ F.reshape(C * N, -1).double().T) for F in (V1, V2, Vw))
    ok["gram"] = bool((G.double() - ref).abs().max() <= 1e-5 * ref.abs().max())
    refg = V1.reshape(C * N, -1).double() @ g1.reshape(N, -1).double().T
    ok["vtg"] = bool((VtG.double() - refg).abs().max() <= 1e-5 * refg.abs().max())

    # (2) DirectionalDampedNewtonComputation(data_parallel=True, factorised) vs the one-process run on the whole batch
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(30, 24), nn.ReLU(), nn.Linear(24, 10)).to(dev)
    Nb = 64 * world
    X, y = torch.rand(Nb, 30, generator=torch.Generator().manual_seed(1)).to(dev), torch.randint(0, 10, (Nb,), generator=torch.Generator().manual_seed(2)).to(dev)

    def run(Xs, ys, **kw):
        m, lossf = extend(model), extend(nn.CrossEntropyLoss())
        comp = vivit_amd.DirectionalDampedNewtonComputation(warn_small_eigvals=0.0, factorised=True, **kw)
        group = {"params": list(m.parameters()), "criterion": top_k_criterion(4, must_exceed=1e-6),
                 "damping": constant_damping(1.0)}
        m.zero_grad()
        with backpack(*comp.get_extensions(), extension_hook=comp.get_extension_hook([group])):
            lossf(m(Xs), ys).backward()
        return comp.get_result(group)

    full = run(X, y)
    lo, hi = rank * 64, (rank + 1) * 64
    shard = run(X[lo:hi], y[lo:hi], data_parallel=True)
    ok["newton"] = all(bool((a - b).abs().max() <= 1e-4 * b.abs().max() + 1e-7) for a, b in zip(shard, full))
    ret[rank] = ok
    dist.destroy_process_group()


def test_data_parallel_on_hip_kernels_world2():
    world = 2
    port = 29500 + (os.getpid() % 1000)
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert r in ret and all(ret[r].values()), dict(ret)


def _worker_band(rank, world, port, ret):
    """The sharded band reduction on the HIP kernels (panel QR, GEMMs, banded solver): two ranks on the one GPU."""
    _one_card_env()
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vivit_amd import distributed as vd, kernels

    dev = torch.device("cuda:0")
    ok = {}
    for n, kind in ((200, "dense"), (1000, "dense"), (2100, "lowrank")):
        g = torch.Generator().manual_seed(n)
        if kind == "dense":
            M = torch.randn(n, n, generator=g)
            S = ((M + M.T) / 2).to(dev)
        else:   # a Gram matrix of rank n / 3: the spectrum this solver is for
            V = torch.randn(n, n // 3, generator=g).to(dev)
            S = V @ V.T
        w_ref = torch.linalg.eigvalsh(S.double().cpu())
        scale = float(w_ref.abs().max())
        w, Z = vd.symeig(S, sharded_reduction=True)
        ok[f"vals{n}"] = bool((w.double().cpu() - w_ref).abs().max() <= 5e-6 * scale)
        Zd, Sd = Z.double(), S.double()
        ok[f"resid{n}"] = bool((Sd @ Zd - Zd * w.double()).abs().max() <= 2e-5 * scale)
        ok[f"orth{n}"] = bool((Zd.T @ Zd - torch.eye(n, dtype=torch.float64, device=dev)).abs().max() <= 2e-5)
        # against the replicated reduction (same eigenvalues to fp32 accuracy; the band itself differs by rounding only)
        w_rep, _ = vd.symeig(S, sharded_reduction=False)
        ok[f"replicated{n}"] = bool((w - w_rep).abs().max() <= 5e-6 * scale)
        if kind == "dense":
            # The band itself, entry by entry, against vivit_sy2sb_f32 (which delays the trailing update over groups of four
            # panels: another rounding order).  Reflector signs are fixed by the data as long as the panels have full rank --
            # in the numerical null space of a low-rank Gram matrix the reflectors follow the rounding noise, and only the
            # spectrum is comparable.
            _, tau1_single, A_single = kernels.sy2sb(S)
            A = S.clone()
            kernels.symeig_prepare_(A)
            vd.sy2sb_sharded_(A)
            band = lambda X: torch.tril(X) - torch.tril(X, -65)   # noqa: E731
            ok[f"band{n}"] = bool((band(A) - band(A_single)).abs().max() <= 2e-4 * scale)
    ret[rank] = ok
    dist.destroy_process_group()


def test_sharded_band_reduction_on_hip_kernels_world2():
    world = 2
    port = 29700 + (os.getpid() % 1000)
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker_band, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        assert r in ret and all(ret[r].values()), dict(ret)


def _worker_rccl(rank, world, port, ret):
    """ONE rank on backend "nccl" (= RCCL) with every collective FORCED (VIVIT_DIST_FORCE_COLLECTIVES=1): the one-GPU box
    executes the code the 8-GPU run takes -- init_process_group("nccl", device_id=...), all_to_all_single(async_op=True) +
    work.wait() ordered against the SYRK stream over several column chunks, the packed all-reduce, all-gathers and the
    per-panel broadcast / all-gather of the sharded band reduction -- through RCCL kernels on device memory (no host
    staging).  Results must equal the single-process kernels (the exchange is an identity here, bit for bit)."""
    os.environ["VIVIT_DIST_FORCE_COLLECTIVES"] = "1"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import vivit_amd
    from helpers import top_k_criterion
    from torch import nn
    from vivit_amd import distributed as vd, kernels
    from vivit_amd.backend import backpack, extend

    ok = {"backend": dist.get_backend() == "nccl", "forced": vd._forced() and vd._active()}
    C, N = 10, 64
    g = torch.Generator().manual_seed(5)
    V1 = torch.randn(C, N, 20500, generator=g).to(dev)        # 20 500 columns: three exchange chunks (8192 + 8192 + 4116)
    V2 = torch.randn(C, N, 40, generator=g).to(dev)           # narrow: all-gather + block rows
    s = torch.randn(C, N, 24, generator=g).to(dev)
    z = torch.randn(N, 50, generator=g).to(dev)
    g1 = torch.randn(N, 20500, generator=g).to(dev)
    acc = vd.BatchShardedGram(C, N, N_grad_local=N)
    acc.add_factor(V1, g1)
    acc.add_factor_rows(V2)
    acc.add_linear(s, z)
    G = acc.finalize().reshape(C * N, C * N)
    VtG = acc.finalize_vtg().reshape(C * N, N)
    torch.cuda.synchronize()
    # the same sums on the single-process kernels, chunk by chunk in the same order: bit-identical
    ref = torch.zeros(C * N, C * N, device=dev)
    refg = torch.zeros(C * N, N, device=dev)
    A1 = V1.reshape(C * N, -1)
    for lo in range(0, 20500, vd.EXCHANGE_CHUNK_COLUMNS):
        hi = min(lo + vd.EXCHANGE_CHUNK_COLUMNS, 20500)
        kernels.gram_syrk(A1[:, lo:hi].contiguous(), out=ref, alpha=1.0, beta=1.0)
        kernels.gemm_nt(A1[:, lo:hi].contiguous(), g1[:, lo:hi].contiguous(), out=refg, alpha=1.0, beta=1.0)
    rows = kernels.gemm_nt(V2.reshape(C * N, -1), V2.reshape(C * N, -1))
    Gz = kernels.gemm_nt(z, z)
    Gs = kernels.gemm_nt(s.reshape(C * N, -1), s.reshape(C * N, -1))
    rows = kernels.gram_hadamard_block(Gz, Gs, C, N, C, N, out=rows, alpha=1.0, beta=1.0)
    ok["gram_bitwise"] = bool(torch.equal(G, ref + rows))
    ok["vtg_bitwise"] = bool(torch.equal(VtG, refg))
    ref64 = sum((F.reshape(C * N, -1).double() @ F.reshape(C * N, -1).double().T)
                for F in (V1, V2, torch.einsum("cno,ni->cnoi", s, z)))
    ok["gram_fp64"] = bool((G.double() - ref64).abs().max() <= 1e-5 * ref64.abs().max())
    ok["symmetric"] = bool(torch.equal(G, G.T))                  # the packed lower triangle travelled, the rest is mirrored
    # back-projection: all-reduce of K P floats
    coef = torch.randn(3, C, N, generator=g).to(dev)
    bp = vd.backproject_sum(coef, V2, acc)
    ok["backproject"] = bool(torch.allclose(bp, coef.reshape(3, -1) @ V2.reshape(C * N, -1), rtol=1e-4, atol=1e-4))
    # eigensolver: row-sharded back-transformations + all-gather, and the sharded band reduction (2 collectives per panel)
    w_ref = torch.linalg.eigvalsh(G.double().cpu())
    scale = float(w_ref.abs().max())
    for tag, kw in (("rows", {"sharded_reduction": False}), ("band", {"sharded_reduction": True})):
        w, Z = vd.symeig(G, **kw)
        ok[f"symeig_{tag}_vals"] = bool((w.double().cpu() - w_ref).abs().max() <= 5e-6 * scale)
        Zd = Z.double()
        ok[f"symeig_{tag}_resid"] = bool((G.double() @ Zd - Zd * w.double()).norm(dim=0).max() <= 1e-5 * scale)
    # the public API with data_parallel=True on the nccl group
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(30, 24), nn.ReLU(), nn.Linear(24, 10)).to(dev)
    X = torch.rand(64, 30, generator=torch.Generator().manual_seed(1)).to(dev)
    y = torch.randint(0, 10, (64,), generator=torch.Generator().manual_seed(2)).to(dev)

    def run(**kw):
        m, lossf = extend(model), extend(nn.CrossEntropyLoss())
        comp = vivit_amd.EighComputation(warn_small_eigvals=0.0, **kw)
        group = {"params": list(m.parameters()), "criterion": top_k_criterion(4, must_exceed=1e-6)}
        m.zero_grad()
        with backpack(comp.get_extension(), extension_hook=comp.get_extension_hook([group])):
            lossf(m(X), y).backward()
        return comp.get_result(group)

    (e0, v0), (e1, v1) = run(), run(data_parallel=True)
    ok["api_evals"] = bool((e0 - e1).abs().max() <= 1e-5 * e0.abs().max())
    ok["api_evecs"] = all(bool((a.abs() - b.abs()).abs().max() <= 1e-3) for a, b in zip(v0, v1))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_collectives_forced_on_one_rank():
    """the RCCL path executes at least once on the one-GPU box (scaling stays unmeasured there).
    Reference accumulation being sharded: vivit/utils/gram.py:104-116."""
    port = 29900 + (os.getpid() % 1000)
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker_rccl, args=(1, port, ret), nprocs=1, join=True)
    assert 0 in ret and all(ret[0].values()), dict(ret)
