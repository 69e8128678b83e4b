"""Selected eigenvectors (vivit_symeig_reduce_f32 + vivit_symeig_select_f32) against fp64 LAPACK on the host:
eigenvalues, sign-free eigenvectors of well-separated eigenvalues, residual and orthonormality for every selection
(incl. numerically multiple eigenvalues, where only the eigenspace is defined), arbitrary order / repeated indices,
both reductions (one-stage below n = 2048, two-stage above), inverse iteration (K <= 256) and the D&C route."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _matrix(n, kind, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "lowrank":      # GGN-like: rank-deficient Gram matrix with a decaying spectrum
        p = n // 2
        V = torch.randn(n, p, generator=g, dtype=torch.float64) * (0.97 ** torch.arange(p, dtype=torch.float64))
        return V @ V.T
    if kind == "clustered":    # five 8-fold eigenvalues on top of a noise floor
        Q, _ = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))
        lam = torch.rand(n, generator=g, dtype=torch.float64) * 1e-3
        for c in range(5):
            lam[n - 8 * (c + 1): n - 8 * c] = 10.0 - c
        return (Q * lam) @ Q.T
    M = torch.randn(n, n, generator=g, dtype=torch.float64)
    return (M + M.T) / 2


def _check(G64, plan, keep, dev, sep_tol=1e-4):
    n = G64.shape[0]
    wref, Zref = np.linalg.eigh(G64.numpy())
    scale = np.abs(wref).max()
    w = plan.evals.cpu().double().numpy()
    assert np.abs(w - wref).max() <= 1e-5 * scale
    Z = plan.select(keep)
    assert Z.shape == (n, len(keep))
    Zd = Z.cpu().double().numpy()
    keep_pos = [k % n for k in keep]
    # residual of every returned pair and orthonormality among distinct selections
    R = G64.numpy() @ Zd - Zd * wref[keep_pos]
    assert np.abs(R).max() <= 3e-5 * scale, np.abs(R).max() / scale
    uniq = sorted(set(keep_pos))
    Zu = Zd[:, [keep_pos.index(k) for k in uniq]]
    assert np.abs(Zu.T @ Zu - np.eye(len(uniq))).max() <= 1e-4
    # sign-free equality where the eigenvalue is isolated (test/linalg/test_eigh.py:147-153 of the reference)
    for col, k in enumerate(keep_pos):
        gap = min(abs(wref[k] - wref[k - 1]) if k > 0 else np.inf, abs(wref[k + 1] - wref[k]) if k + 1 < n else np.inf)
        if gap > sep_tol * scale:
            assert np.abs(np.abs(Zd[:, col]) - np.abs(Zref[:, k])).max() <= 2e-2 * (1e-4 * scale / gap) + 2e-3


@pytest.mark.parametrize("n,kind", [(193, "dense"), (500, "lowrank"), (1000, "clustered"), (2048, "lowrank"),
                                    (3000, "dense"), (4096, "clustered")])
def test_select_topk(n, kind):
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    G64 = _matrix(n, kind, seed=n)
    G = G64.float().to(dev)
    plan = kernels.symeig_reduce(G)
    assert torch.equal(G.cpu(), G64.float()), "input must not be modified without overwrite=True"
    _check(G64.float().double(), plan, list(range(n - 10, n)), dev)
    # a second selection from the same reduction: one vector, unsorted + repeated indices, negative indices
    _check(G64.float().double(), plan, [n - 1], dev)
    _check(G64.float().double(), plan, [n - 3, n - 40, n - 3, -1], dev)
    assert plan.select([]).shape == (n, 0)
    with pytest.raises(IndexError):
        plan.select([n])


@pytest.mark.parametrize("n", [700, 2500])
def test_select_many_uses_divide_and_conquer(n):
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    G64 = _matrix(n, "lowrank", seed=n + 1).float().double()
    plan = kernels.symeig_reduce(G64.float().to(dev), overwrite=True)
    keep = list(range(n - 300, n))          # > 256 vectors
    _check(G64, plan, keep, dev)


def test_select_matches_full_solver():
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    n = 2304
    G = _matrix(n, "lowrank", seed=5).float().to(dev)
    w_full, Z_full = kernels.symeig(G, eigenvectors=True)
    plan = kernels.symeig_reduce(G)
    assert (plan.evals - w_full).abs().max().item() <= 1e-5 * w_full[-1].item()
    keep = list(range(n - 8, n))
    Z = plan.select(keep)
    dots = (Z * Z_full[:, keep]).sum(0).abs()      # isolated top eigenvalues: same vectors up to sign
    assert (1 - dots).max().item() <= 1e-3


def test_select_api_eigh_topk_large_group():
    """EighComputation through the two-phase solver (n = 400 > 192) equals the golden-path semantics: Ge = le."""
    import vivit_amd
    from helpers import FakeModule, top_k_criterion

    dev = torch.device("cuda:0")
    C, N, P = 4, 100, 900
    g = torch.Generator().manual_seed(0)
    V = (torch.randn(C, N, P, generator=g) * (0.98 ** torch.arange(P))).to(dev)
    p = torch.nn.Parameter(torch.zeros(P, device=dev))
    p.sqrt_ggn_exact = V
    from vivit_amd.backend.extensions import _materialised_closures

    comp = vivit_amd.EighComputation()
    setattr(p, comp._savefield, _materialised_closures(V))
    group = {"params": [p], "criterion": top_k_criterion(6)}
    mod = FakeModule([p], N)
    mod.input0 = torch.zeros(N, 1, device=dev)
    comp.get_extension_hook([group])(mod)
    evals, (evecs,) = comp.get_result(group)
    A = V.reshape(C * N, P).double()
    H = (A.T @ A).cpu()
    wref = np.linalg.eigvalsh(H.numpy())[-6:]
    np.testing.assert_allclose(evals.cpu().double().numpy(), wref, rtol=1e-4, atol=5e-6)
    E = evecs.cpu().double()
    np.testing.assert_allclose((E @ H).numpy(), (evals.cpu().double()[:, None] * E).numpy(), rtol=1e-3, atol=2e-4 * wref.max())
