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
