"""BASELINE config 1 on the HIP path, end to end: 3-layer MLP 784-512-10, batch 128 (n = N C = 1 280, P = 407 050),
``extend`` + ``with backpack(...)`` + ``EigvalshComputation`` / ``EighComputation`` exactly as the reference's
docs/examples/basic_usage/example_eigvalsh.py:55-83 drives it -- the one BASELINE workload where the CPU oracle fits
completely, so ALL 1 280 eigenvalues are compared (reference tests: test/linalg/test_eigvalsh.py:27-63,
test/linalg/test_eigh.py:28-155).  The oracle side runs in fp64 on its own brute-force factors
(oracle.sqrt_ggn_factors: per-sample Jacobians x loss-Hessian square root), i.e. it shares nothing with the product but
the model weights and the data.

Plus the same MLP at batch 512 (n = 5 120: the chunked 256-tile bf16-pipe SYRK on REAL materialised factors, 8.3 GB):
spectrum of the HIP path against fp64 ``eigvalsh`` of the fp64 Gram matrix of the very same factors.
"""
import numpy as np
import pytest
import torch
from torch import nn

import vivit_amd
from oracle import vivit_oracle as oracle
from vivit_amd.backend import SqrtGGNExact, backpack, extend

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

DIMS = (784, 512, 10)


def mlp(seed=0):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(DIMS[0], DIMS[1]), nn.ReLU(), nn.Linear(DIMS[1], DIMS[2]))


def data(N, seed=0):
    g = torch.Generator().manual_seed(seed + 1)
    return torch.rand(N, DIMS[0], generator=g), torch.randint(0, DIMS[2], (N,), generator=g)


def run_backward(model, X, y, extensions, hook=None):
    model, lossf = extend(model), extend(nn.CrossEntropyLoss())
    model.zero_grad()
    loss = lossf(model(X), y)
    with backpack(*extensions, extension_hook=hook):
        loss.backward()


@pytest.fixture(scope="module")
def oracle_side():
    """fp64 oracle: factors, Gram matrix (vivit/utils/gram.py:206-232), spectrum, top eigenvectors in parameter space."""
    N = 128
    X, y = data(N)
    ref = mlp().double()
    S = oracle.loss_hessian_sqrt_exact(ref(X.double()).detach(), "ce")
    V = oracle.sqrt_ggn_factors(ref, X.double(), S)                      # [C, N, *param] per parameter, fp64
    grams = [oracle.pairwise_dot(v, start_dim=2, flatten=False) for v in V]
    evals = oracle.eigvalsh_group(grams, N, None)
    return {"N": N, "X": X, "y": y, "V": V, "grams": grams, "evals": evals}


def test_config1_eigvalsh_full_spectrum(oracle_side):
    """All 1 280 eigenvalues of the Gram matrix built and decomposed by the HIP kernels through the public API."""
    dev = torch.device("cuda:0")
    N, X, y = oracle_side["N"], oracle_side["X"].to(dev), oracle_side["y"].to(dev)
    model = mlp().to(dev)
    for flavour in ("materialised", "factorised"):
        comp = vivit_amd.EigvalshComputation()
        group = {"params": list(model.parameters())}
        ext = comp.get_extension() if flavour == "factorised" else None
        if flavour == "materialised":   # BackPACK's own SqrtGGNExact layout -> the generic K1 path (4 SYRKs, K = 401 408)
            from vivit_amd.extensions.hooks import GramSqrtGGNExact

            hook = GramSqrtGGNExact(free_sqrt_ggn=True)
            run_backward(model, X, y, [SqrtGGNExact()], hook)
            from vivit_amd import kernels

            w, _ = kernels.symeig(hook.get_result().clone(), eigenvectors=False)
        else:                           # ViViTGGNExact closures (Linear weights stay factorised, linear.py:41-81)
            run_backward(model, X, y, [ext], comp.get_extension_hook([group]))
            w = comp.get_result(group)
        w = w.cpu().double().numpy()
        ref = oracle_side["evals"].numpy()
        assert w.shape == ref.shape == (N * DIMS[2],)
        lam = ref[-1]
        # every eigenvalue at rtol 1e-4 / atol 1e-5 lambda_max ...
        np.testing.assert_allclose(w, ref, rtol=1e-4, atol=1e-5 * lam, err_msg=flavour)
        # ... and BASELINE's "eigenvalues within 1e-5 rel-err", scoped like test_eigvalsh.py:55-60 (to lambda_max)
        assert np.abs(w - ref).max() <= 1e-5 * lam, (flavour, np.abs(w - ref).max() / lam)
        # rank(G) <= N (C - 1): the bottom N eigenvalues are zero
        assert np.abs(w[:N]).max() <= 1e-5 * lam


def test_config1_eigh_keep_all(oracle_side):
    """``EighComputation`` with the reference tests' ``keep_nonzero`` criterion (test/linalg/settings.py:35-44):
    eigenvalues, orthonormality, G e = lambda e against the ORACLE's factors, and the leading eigenvectors sign-free."""
    dev = torch.device("cuda:0")
    N, X, y = oracle_side["N"], oracle_side["X"].to(dev), oracle_side["y"].to(dev)
    model = mlp().to(dev)
    ref_w = oracle_side["evals"]
    lam = float(ref_w[-1])

    def keep_nonzero(evals):
        return [i for i in range(evals.numel()) if float(evals[i]) > 1e-4 * float(evals[-1])]

    comp = vivit_amd.EighComputation(warn_small_eigvals=0.0)
    group = {"params": list(model.parameters()), "criterion": keep_nonzero}
    run_backward(model, X, y, [comp.get_extension()], comp.get_extension_hook([group]))
    evals, evecs = comp.get_result(group)
    K = evals.numel()
    keep_ref = [i for i in range(ref_w.numel()) if float(ref_w[i]) > 1e-4 * lam]
    assert abs(K - len(keep_ref)) <= 2   # eigenvalues sitting on the threshold may fall either side
    Kc = min(K, len(keep_ref))
    np.testing.assert_allclose(evals.cpu().double().numpy()[-Kc:], ref_w.numpy()[-Kc:], rtol=1e-4, atol=1e-5 * lam)
    E = torch.cat([e.flatten(1) for e in evecs], 1)                       # [K, P] on the device
    assert E.shape[1] == sum(p.numel() for p in model.parameters())
    overlap = (E.double() @ E.double().T).cpu().numpy()
    # test_eigh.py:135-141 asks for atol 2e-4 on problems with a handful of O(1) eigenvalues.  Here the kept spectrum spans
    # four decades.  A backward-stable fp32 eigensolver returns pairs with G e~ = l e~ + r, |r| = O(eps lambda_max), and
    # e_i . e_j = e~_i^T (V V^T) e~_j / sqrt(l_i l_j), so the deviation from orthonormality carries the amplification
    # lambda_max / sqrt(l_i l_j) (1e4 for two directions at the criterion's threshold): measured 4.7e-7 = 4 eps times it.
    # LAPACK's ssyevd behind the reference's Tensor.symeig obeys the same bound.
    ev = evals.cpu().double().numpy()
    dev_ = np.abs(overlap - np.eye(K))
    amp = lam / np.sqrt(np.outer(ev, ev))
    print("orthonormality: max deviation", dev_.max(), "max deviation / amplification", (dev_ / amp).max())
    assert (dev_ <= np.maximum(2e-4, 6e-7 * amp)).all(), (dev_.max(), (dev_ / amp).max())
    strong = ev >= 3e-3 * lam                                             # amplification <= 333: the reference's atol holds
    np.testing.assert_allclose(overlap[np.ix_(strong, strong)], np.eye(int(strong.sum())), atol=2e-4)
    # scaling property on a spread of directions, with the oracle's fp64 factors: (V^T V) e = lambda e
    Vflat = torch.cat([v.reshape(N * DIMS[2], -1) for v in oracle_side["V"]], 1)   # [n, P] fp64 (CPU)
    pick = sorted(set([K - 1, K - 2, K - 3, K - 10, K // 2, K // 4, 5, 0]))
    Es = E[pick].cpu().double()
    GE = (Es @ Vflat.T) @ Vflat
    lam_s = evals[pick].cpu().double()
    np.testing.assert_allclose(GE.numpy(), (lam_s[:, None] * Es).numpy(), rtol=5e-4, atol=1e-5 * lam / 10)
    # leading eigenvectors against the oracle's (vivit/linalg/eigh.py:222-275 restated), up to sign
    top = 5
    ref_evals, ref_evecs = oracle.eigh_group(
        oracle_side["grams"], [lambda m, v=v: oracle.Vmp(v, m, 2) for v in oracle_side["V"]],
        lambda ev: list(range(ev.numel() - top, ev.numel())), N, None)
    R = torch.cat([e.flatten(1) for e in ref_evecs], 1)
    gaps = (ref_evals[1:] - ref_evals[:-1]).min() / lam
    assert gaps > 1e-3, "test assumes separated leading eigenvalues"
    np.testing.assert_allclose(E[-top:].cpu().double().abs().numpy(), R.abs().numpy(), rtol=2e-2, atol=2e-4)


def test_mlp_b512_spectrum_vs_fp64_gram():
    """n = 5 120, P = 407 050, real factors from the backend: HIP Gram + symeig against fp64 all the way."""
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    N = 512
    X, y = data(N, seed=3)
    model = mlp(seed=3).to(dev)
    run_backward(model, X.to(dev), y.to(dev), [SqrtGGNExact()])
    n = N * DIMS[2]
    facs = [p.sqrt_ggn_exact.reshape(n, -1) for p in model.parameters()]
    assert sum(f.shape[1] for f in facs) == 407050
    G = None
    for f in facs:
        G = kernels.gram_syrk(f, out=G, alpha=1.0, beta=0.0 if G is None else 1.0)
    G64 = torch.zeros((n, n), dtype=torch.float64, device=dev)
    for f in facs:
        for c0 in range(0, f.shape[1], 32768):
            blk = f[:, c0:c0 + 32768].double()
            G64.addmm_(blk, blk.T)
    # Gram entries: error relative to sqrt(G_ii G_jj), the scale of an entry's terms
    dscale = G64.diagonal().sqrt()
    ent = ((G.double() - G64).abs() / (dscale[:, None] * dscale[None, :])).max().item()
    assert ent <= 3e-6, ent
    ref = np.linalg.eigvalsh(G64.cpu().numpy())
    w, Z = kernels.symeig(G, eigenvectors=True)
    w_only, _ = kernels.symeig(G, eigenvectors=False)
    lam = ref[-1]
    for got in (w, w_only):
        err = np.abs(got.cpu().double().numpy() - ref).max() / lam
        assert err <= 1e-5, err
    # eigenvectors: residual and orthonormality against the fp64 Gram matrix
    Zd = Z.double()
    res = (G64 @ Zd - Zd * w.double()[None, :]).abs().max().item() / lam
    orth = (Zd.T @ Zd - torch.eye(n, dtype=torch.float64, device=dev)).abs().max().item()
    assert res <= 2e-5 and orth <= 1e-4, (res, orth)
