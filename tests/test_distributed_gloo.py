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
    from helpers import OracleBackend, set_kernel_backend
    from vivit_amd import kernels

    set_kernel_backend(OracleBackend())


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

    # (6) unequal shards are rejected -- by the check itself and by the accumulator's constructor (a ragged last batch
    # would otherwise give mismatched collective sizes: a hang or corruption under RCCL)
    try:
        vd.check_equal_shards(Ng + rank)
        ok["unequal_raises"] = False
    except ValueError:
        ok["unequal_raises"] = True
    try:
        vd.BatchShardedGram(C, Ng + rank)
        ok["unequal_ctor_raises"] = False
    except ValueError:
        ok["unequal_ctor_raises"] = True
    ok["equal_ok"] = vd.check_equal_shards(Ng) == N

    # (7) the headline's aspect (C = 10, a wide parameter and three small ones), exchanged in several column chunks with
    # a ragged last one: chunk j + 1 is in flight while chunk j is multiplied; packed lower-triangle all-reduce
    vd.EXCHANGE_CHUNK_COLUMNS = 37
    Cw, Pw = 10, 40 * world + 3
    Vw1 = torch.randn(Cw, N, Pw, generator=g)
    Vw2 = torch.randn(Cw, N, 4, generator=g)
    gw1 = torch.randn(N, Pw, generator=g)
    acc = vd.BatchShardedGram(Cw, Ng, N_grad_local=Ng)
    acc.add_factor(Vw1[:, lo:hi], gw1[lo:hi])
    acc.add_factor(Vw2[:, lo:hi], torch.zeros(Ng, 4))
    Gw = acc.finalize()
    VtGw = acc.finalize_vtg()
    ok["chunked_gram"] = torch.allclose(Gw, oracle.compute_gram_mat([Vw1, Vw2], start_dim=2, flatten=False), rtol=1e-5, atol=1e-4)
    ok["chunked_vtg"] = torch.allclose(VtGw, oracle.partial_contract(Vw1, gw1, (2, 1)), rtol=1e-5, atol=1e-4)
    Gf = Gw.reshape(Cw * N, Cw * N)
    ok["chunked_symmetric"] = torch.equal(Gf, Gf.T)
    flat = Gf.reshape(-1)
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok["chunked_replicated"] = all(torch.equal(gathered[0], t) for t in gathered)
    ret[rank] = {k: bool(v) for k, v in ok.items()}
    dist.destroy_process_group()


def _worker_dp_newton(rank, world, port, ret):
    """DirectionalDampedNewtonComputation(data_parallel=True): every rank back-propagates ITS batch shard; the step
    (and gammas / lambdas) must equal the single-process result on the whole batch (oracle on autograd factors)."""
    _setup(rank, world, port)
    import vivit_amd
    from helpers import constant_damping, top_k_criterion
    from oracle import vivit_oracle as oracle
    from torch import nn
    from vivit_amd.backend import backpack, extend

    ok = {}
    for factorised in (False, True):
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(7, 6), nn.ReLU(), nn.Linear(6, 5))
        Ng = 3
        N = Ng * world
        X, y = torch.rand(N, 7), torch.randint(0, 5, (N,))
        ref = nn.Sequential(nn.Linear(7, 6), nn.ReLU(), nn.Linear(6, 5))
        ref.load_state_dict(model.state_dict())
        S = oracle.loss_hessian_sqrt_exact(ref(X).detach(), "ce")
        V_ref = oracle.sqrt_ggn_factors(ref, X, S, None)
        g_ref = oracle.batch_grads(ref, X, y, nn.CrossEntropyLoss(), None)
        crit = top_k_criterion(3, must_exceed=1e-4)
        ref_steps = oracle.damped_newton_group(V_ref, g_ref, crit, constant_damping(1.0), N)
        ref_gam, ref_lam = oracle.directional_derivatives_group(V_ref, g_ref, crit, N)

        lo, hi = rank * Ng, (rank + 1) * Ng
        for cls, check in ((vivit_amd.DirectionalDampedNewtonComputation, "step"),
                           (vivit_amd.DirectionalDerivativesComputation, "dirs")):
            m, lossf = extend(model), extend(nn.CrossEntropyLoss())
            comp = cls(warn_small_eigvals=0.0, factorised=factorised, data_parallel=True)
            group = {"params": list(m.parameters()), "criterion": crit, "damping": constant_damping(1.0)}
            m.zero_grad()
            loss = lossf(m(X[lo:hi]), y[lo:hi])         # the rank's own shard only
            with backpack(*comp.get_extensions(), extension_hook=comp.get_extension_hook([group])):
                loss.backward()
            res = comp.get_result(group)
            if check == "step":
                good = all(torch.allclose(s_, r_, rtol=1e-3, atol=2e-5 * max(r_.abs().max().item(), 1e-2))
                           for s_, r_ in zip(res, ref_steps))
                flat = torch.cat([s_.reshape(-1) for s_ in res])
                gathered = [torch.empty_like(flat) for _ in range(world)]
                dist.all_gather(gathered, flat)
                good = good and all(torch.equal(gathered[0], t) for t in gathered)   # replicated step
            else:
                gam, lam = res
                good = (gam.shape == ref_gam.shape
                        and torch.allclose(gam.abs(), ref_gam.abs(), rtol=1e-3, atol=1e-4 * ref_gam.abs().max().item())
                        and torch.allclose(lam, ref_lam, rtol=1e-3, atol=1e-5 * ref_lam.abs().max().item()))
            ok[f"{check}_{'fact' if factorised else 'mat'}"] = bool(good)
    ret[rank] = ok
    dist.destroy_process_group()


def _worker_dp_linalg(rank, world, port, ret):
    """EigvalshComputation / EighComputation(data_parallel=True) on batch shards == dense GGN of the whole batch."""
    _setup(rank, world, port)
    import vivit_amd
    from oracle import vivit_oracle as oracle
    from torch import nn
    from vivit_amd.backend import backpack, extend

    torch.manual_seed(0)
    model = nn.Sequential(nn.Conv2d(2, 2, 2), nn.Flatten(), nn.Tanh(), nn.Linear(18, 4))
    Ng = 3
    N = Ng * world
    X, y = torch.rand(N, 2, 4, 4), torch.randint(0, 4, (N,))
    ref = nn.Sequential(nn.Conv2d(2, 2, 2), nn.Flatten(), nn.Tanh(), nn.Linear(18, 4))
    ref.load_state_dict(model.state_dict())
    ggn = oracle.dense_ggn(ref.double(), X.double(), "ce")
    w_ref, Q_ref = torch.linalg.eigh(ggn)
    lo, hi = rank * Ng, (rank + 1) * Ng
    ok = {}

    def run(comp, group):
        m, lossf = extend(model), extend(nn.CrossEntropyLoss())
        m.zero_grad()
        loss = lossf(m(X[lo:hi]), y[lo:hi])
        with backpack(comp.get_extension(), extension_hook=comp.get_extension_hook([group])):
            loss.backward()
        return comp.get_result(group)

    params = list(model.parameters())
    P = sum(p.numel() for p in params)
    ev = run(vivit_amd.EigvalshComputation(data_parallel=True), {"params": params})
    k = min(4 * N, P)
    ok["eigvalsh"] = ev.shape == (4 * N,) and torch.allclose(ev[-k:].double(), w_ref[-k:], rtol=1e-4, atol=5e-6)
    keep = lambda evals: [i for i, e in enumerate(evals) if e > 1e-4]  # noqa: E731
    evals, evecs = run(vivit_amd.EighComputation(data_parallel=True, warn_small_eigvals=0.0), {"params": params, "criterion": keep})
    K = evals.numel()
    E = torch.cat([e.reshape(K, -1) for e in evecs], 1).double()
    ok["eigh_vals"] = torch.allclose(evals.double(), w_ref[-K:], rtol=5e-4, atol=1e-5)
    ok["eigh_resid"] = torch.allclose(E @ ggn, evals.double()[:, None] * E, rtol=1e-3, atol=2e-4)
    ok["eigh_orth"] = torch.allclose(E @ E.T, torch.eye(K, dtype=torch.float64), atol=2e-3)
    ret[rank] = {k_: bool(v) for k_, v in ok.items()}
    dist.destroy_process_group()


def _worker_sharded_band(rank, world, port, ret):
    """The eigensolver with the full -> band reduction sharded by block rows: same spectrum and valid
    eigenvectors as the plain solve, bit-identical on every rank, band equal to the single-process reduction."""
    _setup(rank, world, port)
    from vivit_amd import distributed as vd, kernels

    ok = {}
    solo = [dist.new_group([r]) for r in range(world)][rank]   # (collective: every rank creates every group, same order)
    for n in (130, 200, 333):   # 3, 4 and 6 row blocks of 64 (the last one partial)
        g = torch.Generator().manual_seed(n)
        M = torch.randn(n, n, generator=g)
        S = (M + M.T) / 2
        w_ref = torch.linalg.eigvalsh(S.double())
        scale = float(w_ref.abs().max())
        w, Z = vd.symeig(S, sharded_reduction=True)
        ok[f"vals{n}"] = bool((w.double() - w_ref).abs().max() <= 2e-5 * scale)
        Zd = Z.double()
        ok[f"resid{n}"] = bool((S.double() @ Zd - Zd * w.double()).abs().max() <= 5e-5 * scale)
        ok[f"orth{n}"] = bool((Zd.T @ Zd - torch.eye(n, dtype=torch.float64)).abs().max() <= 5e-5)
        both = [torch.empty_like(Z) for _ in range(world)]
        dist.all_gather(both, Z.contiguous())
        ok[f"same{n}"] = all(torch.equal(both[0], b) for b in both)
        # the band itself against a single-process reduction with the same panel kernel (world of one)
        A = S.clone()
        kernels.symeig_prepare_(A)
        tau1 = vd.sy2sb_sharded_(A, None if world == 1 else dist.group.WORLD)
        A1 = S.clone()
        kernels.symeig_prepare_(A1)
        tau1_solo = vd.sy2sb_sharded_(A1, solo)
        band = lambda X: torch.tril(X) - torch.tril(X, -65)   # noqa: E731
        ok[f"band{n}"] = bool((band(A) - band(A1)).abs().max() <= 2e-5 * scale) and bool((tau1 - tau1_solo).abs().max() <= 1e-5)
        # one collective per panel on the critical path (the all-gather of P); the next block row travels beside it
        nblk = -(-n // 64)
        vd.sy2sb_sharded_(S.clone(), None if world == 1 else dist.group.WORLD)
        ok[f"count{n}"] = vd.LAST_SHARDED_COLLECTIVES == {"all_gather": nblk - 1, "broadcast": nblk, "total": 2 * nblk - 1}
    # the default policy: sharded from SHARDED_BAND_MIN_RANKS ranks and SHARDED_BAND_MIN_N rows on
    vd.LAST_SHARDED_COLLECTIVES.clear()
    S = torch.randn(200, 200, generator=torch.Generator().manual_seed(1))
    S = (S + S.T) / 2
    vd.symeig(S)                                                   # 200 < 8192: replicated reduction
    ok["default_small_replicated"] = not vd.LAST_SHARDED_COLLECTIVES
    vd.SHARDED_BAND_MIN_N, vd.SHARDED_BAND_MIN_RANKS = 192, world
    w, _ = vd.symeig(S)
    ok["default_sharded"] = vd.LAST_SHARDED_COLLECTIVES.get("all_gather") == 3 and bool(
        (w.double() - torch.linalg.eigvalsh(S.double())).abs().max() <= 2e-5 * float(S.abs().max()) * 20)
    ret[rank] = ok
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


def test_data_parallel_newton_step_world2():
    ret = _run(_worker_dp_newton, 2)
    for r in range(2):
        assert r in ret and all(ret[r].values()), ret


def test_data_parallel_linalg_world2():
    ret = _run(_worker_dp_linalg, 2)
    for r in range(2):
        assert r in ret and all(ret[r].values()), ret


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_band_reduction(world):
    ret = _run(_worker_sharded_band, world)
    for r in range(world):
        assert r in ret and all(ret[r].values()), ret
