"""Pin the CPU oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Runs without a GPU."""
import numpy as np
import pytest
import torch

from helpers import BIG_CASES, CASES, constant_damping, golden_factors, load_golden, top_k_criterion
from oracle import vivit_oracle as oracle


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("case", CASES)
def test_contractions(case):
    g = load_golden(case)
    V, G = golden_factors(g)
    close(oracle.compute_gram_mat(V, 2), g["gram_flat"])
    close(oracle.gram_sqrt_ggn(V), g["gram_hook"])
    close(oracle.partial_contract(V[0], G[0], (2, 1)), g["V_t_g0"])
    close(oracle.Vmp(V[0], torch.from_numpy(g["mat"]), 2), g["Vmp0"])
    close(oracle.mVp(V[0], torch.from_numpy(g["pmat"]), 2), g["mVp0"])
    close(oracle.gram_batch_grad([x.clone() for x in G], center=False), g["gram_batch_grad"])
    close(oracle.gram_batch_grad([x.clone() for x in G], center=True), g["gram_batch_grad_centered"])


@pytest.mark.parametrize("case", CASES)
def test_eigvalsh_and_eigh(case):
    g = load_golden(case)
    V, _ = golden_factors(g)
    N, N_total = int(g["N"]), int(g["N_total"])
    sub = None if N == N_total else list(range(N))
    grams = [oracle.pairwise_dot(v, 2, False) for v in V]
    scale = np.abs(g["eigvalsh_one_0"]).max()
    close(oracle.eigvalsh_group(grams, N_total, sub), g["eigvalsh_one_0"], atol=1e-5 * scale)
    for i, gr in enumerate(grams):
        ref = g[f"eigvalsh_per_param_{i}"]
        close(oracle.eigvalsh_group([gr], N_total, sub), ref, atol=1e-5 * np.abs(ref).max())
    evals, evecs = oracle.eigh_group(
        grams, [lambda m, v=v: oracle.Vmp(v, m, 2) for v in V], top_k_criterion(int(g["k"])), N_total, sub
    )
    close(evals, g["eigh_evals"], rtol=1e-4, atol=1e-5 * scale)
    for i, e in enumerate(evecs):
        # eigenvectors are defined up to sign (reference test/linalg/test_eigh.py:147-153)
        close(e.abs(), np.abs(g[f"eigh_evecs{i}"]), rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("case", CASES)
def test_directional_and_newton(case):
    g = load_golden(case)
    V, G = golden_factors(g)
    N_total = int(g["N_total"])
    crit = top_k_criterion(int(g["k"]))
    gam, lam = oracle.directional_derivatives_group(V, G, crit, N_total)
    close(gam.abs(), np.abs(g["gammas"]), rtol=1e-4, atol=1e-4 * np.abs(g["gammas"]).max())
    close(lam, g["lambdas"], rtol=1e-4, atol=1e-5 * np.abs(g["lambdas"]).max())
    steps = oracle.damped_newton_group(V, G, crit, constant_damping(1.0), N_total)
    for i, s in enumerate(steps):
        ref = g[f"newton{i}"]
        close(s, ref, rtol=1e-4, atol=1e-5 * max(np.abs(ref).max(), 1e-3))


@pytest.mark.parametrize("case", BIG_CASES)
def test_two_stage_size_against_the_reference(case):
    """n = 2560 (the size class of the headline's two-stage solver): the oracle against the imported reference's
    EigvalshComputation / EighComputation top-10 / gammas, lambdas / damped Newton step on seeded factors."""
    g = load_golden(case)
    V, G = golden_factors(g)
    N = int(g["N"])
    crit = top_k_criterion(int(g["k"]))
    grams = [oracle.pairwise_dot(v, 2, False) for v in V]
    ref = g["eigvalsh_one_0"]
    scale = np.abs(ref).max()
    close(oracle.eigvalsh_group(grams, N, None), ref, rtol=1e-4, atol=1e-5 * scale)
    evals, evecs = oracle.eigh_group(grams, [lambda m, v=v: oracle.Vmp(v, m, 2) for v in V], crit, N, None)
    close(evals, g["eigh_evals"], rtol=1e-4, atol=1e-5 * scale)
    for i, e in enumerate(evecs):
        close(e.abs(), np.abs(g[f"eigh_evecs{i}"]), rtol=2e-2, atol=2e-3)
    gam, lam = oracle.directional_derivatives_group(V, G, crit, N)
    close(gam.abs(), np.abs(g["gammas"]), rtol=1e-4, atol=1e-4 * np.abs(g["gammas"]).max())
    close(lam, g["lambdas"], rtol=1e-4, atol=1e-5 * np.abs(g["lambdas"]).max())
    for i, s in enumerate(oracle.damped_newton_group(V, G, crit, constant_damping(1.0), N)):
        close(s, g[f"newton{i}"], rtol=1e-4, atol=1e-5 * max(np.abs(g[f"newton{i}"]).max(), 1e-3))


def test_eig_utils():
    g = load_golden("eig_utils")
    for name in ["T1", "T2", "T3"]:
        T = torch.from_numpy(g[name])
        for shift in [0.0, 0.1, 1.0, 10.0]:
            w, _ = oracle.symeig_psd(T.clone(), eigenvectors=True, shift=shift)
            close(w, g[f"{name}_psd_w_{shift}"], rtol=1e-5, atol=1e-5)
        w, v = oracle.symeig(T.clone(), eigenvectors=True)
        close(w, g[f"{name}_symeig_w"], rtol=1e-5, atol=1e-6)
        assert (v.shape[1] if v.numel() else 0) == int(g[f"{name}_symeig_nvec"])
    inp = torch.tensor([[1.0, 1.0], [2.0, 2.0], [3.0, 4.0]])
    close(oracle.shift_diag(inp, 0.1), g["shift_nonsquare"])


@pytest.mark.parametrize("loss", ["ce", "mse"])
def test_factor_oracle_properties(loss):
    """V V^T == dense GGN and Gram spectrum == GGN spectrum
    (test/extensions/secondorder/vivit/test_vivit_ggn.py:22-76, test_gram_sqrt_ggn.py:17-56)."""
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(7, 6), torch.nn.Sigmoid(), torch.nn.Linear(6, 5)).double()
    X = torch.rand(3, 7, dtype=torch.float64)
    out = model(X).detach()
    S = oracle.loss_hessian_sqrt_exact(out, loss)
    V = oracle.sqrt_ggn_factors(model, X, S)
    Vflat = torch.cat([v.flatten(2) for v in V], dim=2).flatten(0, 1)  # [CN, P]
    ggn = oracle.dense_ggn(model, X, loss)
    np.testing.assert_allclose((Vflat.T @ Vflat).numpy(), ggn.numpy(), rtol=1e-9, atol=1e-12)
    gram_w = torch.linalg.eigvalsh(oracle.gram_sqrt_ggn(V))
    ggn_w = torch.linalg.eigvalsh(ggn)
    k = min(gram_w.numel(), ggn_w.numel())
    np.testing.assert_allclose(gram_w[-k:].numpy(), ggn_w[-k:].numpy(), rtol=1e-8, atol=1e-12)


def test_mc_mse_factor_oracle_is_unbiased():
    """The sampled square root of the MSE loss Hessian (oracle.loss_hessian_sqrt_mc_mse; the reference gets it from
    BackPACK's SqrtGGNMSELoss, vivit/extensions/secondorder/vivit/__init__.py:84-86,155-181): sum_m S_m S_m^T must
    converge to the exact Hessian 2 / (N C) I, i.e. to the exact factor's S S^T, and with orthonormal draws it must
    reproduce it exactly."""
    N, C, M = 4, 5, 40000
    gen = torch.Generator().manual_seed(0)
    eps = torch.randn(M, N, C, generator=gen, dtype=torch.float64)
    S = oracle.loss_hessian_sqrt_mc_mse(eps)
    H = torch.einsum("mnc,mnd->ncd", S, S)
    S_exact = oracle.loss_hessian_sqrt_exact(torch.zeros(N, C, dtype=torch.float64), "mse")
    H_exact = torch.einsum("vnc,vnd->ncd", S_exact, S_exact)
    np.testing.assert_allclose(H.numpy(), H_exact.numpy(), atol=0.03 * 2.0 / (N * C))
    # M = C draws that are sqrt(C) times an orthogonal matrix per sample: the sampled factor IS a square root
    Qm = torch.linalg.qr(torch.randn(C, C, generator=gen, dtype=torch.float64))[0] * C ** 0.5
    eps = Qm.unsqueeze(1).expand(C, N, C)
    S = oracle.loss_hessian_sqrt_mc_mse(eps)
    np.testing.assert_allclose(torch.einsum("mnc,mnd->ncd", S, S).numpy(), H_exact.numpy(), rtol=1e-12, atol=1e-15)
