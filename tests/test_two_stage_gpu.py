"""Stages of the two-stage tridiagonalisation (band reduction + bulge chasing) on the GPU."""
import numpy as np
import pytest
import scipy.linalg
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
DEV = "cuda:0"
NB = 64


def to_rowband(band):
    """Full symmetric banded matrix -> [n, 2*NB+1] row-band layout AB[i][j-i+2NB] = A[i][j]."""
    n = band.shape[0]
    AB = np.zeros((n, 2 * NB + 1), np.float32)
    for i in range(n):
        lo = max(0, i - NB)
        AB[i, lo - i + 2 * NB : 2 * NB + 1] = band[i, lo : i + 1]
    return AB


@pytest.mark.parametrize("n", [3, 10, 64, 65, 66, 130, 200, 513, 1000])
def test_sb2st_eigenvalues_and_q2(n):
    from vivit_amd import kernels

    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    M = (M + M.T) / 2
    band = np.triu(np.tril(M, NB), -NB).astype(np.float32)
    ref = np.linalg.eigvalsh(band.astype(np.float64))
    d, e, R2, tau2 = kernels.sb2st(torch.from_numpy(to_rowband(band)).to(DEV))
    d, e = d.cpu().double().numpy(), e.cpu().double().numpy()
    w = scipy.linalg.eigvalsh_tridiagonal(d, e) if n > 1 else d
    scale = np.abs(ref).max()
    assert np.abs(w - ref).max() <= 5e-6 * scale
    if n <= 200:
        # band = Q2 T Q2^T with Q2 = product of the stored reflectors in generation order
        R2h, t2 = R2.cpu().double().numpy(), tau2.cpu().double().numpy()
        Q = np.eye(n)
        for s in range(n - 2):
            k = 0
            while s + 1 + k * NB < n:
                c0 = s + 1 + k * NB
                L = min(NB, n - c0)
                v = np.zeros(n)
                v[c0 : c0 + L] = R2h[s, c0 : c0 + L]
                Q = Q - t2[s, k] * np.outer(Q @ v, v)
                k += 1
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        assert np.abs(Q @ T @ Q.T - band).max() <= 2e-5 * scale
        assert np.abs(Q.T @ Q - np.eye(n)).max() <= 1e-5
