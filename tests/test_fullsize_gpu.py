"""BASELINE-size (n = 40 960) run of Gram build + symeig, checked through size-independent properties
(the oracle cannot run at this size in seconds): ascending order, trace and Frobenius identities,
orthonormality, the eigen-residual, agreement of the two solver flavours.  torch matmuls on the device are
used only as the checker."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


@pytest.mark.parametrize("n,p", [(8192, 4096), (40960, 6144)])
def test_gram_symeig_properties_fullsize(n, p):
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info()
    if free < 6 * n * n * 4 + n * p * 4 + (8 << 30):
        pytest.skip("not enough free HBM for the full-size property test")
    g = torch.Generator(device=dev).manual_seed(n)
    # decaying column scales: a GGN-like spectrum with a numerically rank-deficient tail (rank <= p < n)
    V = torch.randn(n, p, device=dev, generator=g) * (0.999 ** torch.arange(p, device=dev))
    G = kernels.gram_syrk(V)
    del V
    assert torch.equal(G, G.T)
    w_only, _ = kernels.symeig(G, eigenvectors=False)
    w, Z = kernels.symeig(G, eigenvectors=True)
    lam_max = w[-1].item()
    assert lam_max > 0
    assert bool((w[1:] >= w[:-1]).all()), "eigenvalues must be ascending"
    assert (w - w_only).abs().max().item() <= 1e-5 * lam_max          # D&C vs bisection
    assert abs(w.double().sum().item() - G.diagonal().double().sum().item()) <= 1e-5 * n ** 0.5 * lam_max
    fro2 = (G.double() ** 2).sum().item() if n <= 8192 else sum((G[i : i + 4096].double() ** 2).sum().item() for i in range(0, n, 4096))
    assert abs((w.double() ** 2).sum().item() - fro2) <= 1e-4 * fro2
    # orthonormality and residual, in row blocks of Z^T to bound the checker's memory
    worst_orth, worst_res = 0.0, 0.0
    B = 4096
    for i in range(0, n, B):
        Zi = Z[:, i : i + B]                                  # eigenvectors i .. i+B
        gram = Z.T @ Zi                                       # [n, B]
        gram[i : i + Zi.shape[1]] -= torch.eye(Zi.shape[1], device=dev)
        worst_orth = max(worst_orth, gram.abs().max().item())
        res = G @ Zi - Zi * w[i : i + B]
        worst_res = max(worst_res, res.abs().max().item())
        del gram, res
    assert worst_orth <= 1e-4, worst_orth
    assert worst_res <= 3e-5 * lam_max, worst_res / lam_max
