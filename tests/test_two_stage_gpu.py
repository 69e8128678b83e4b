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


def from_rowband(AB):
    n = AB.shape[0]
    band = np.zeros((n, n))
    for i in range(n):
        lo = max(0, i - NB)
        band[i, lo : i + 1] = AB[i, lo - i + 2 * NB : 2 * NB + 1]
    return band + np.tril(band, -1).T


@pytest.mark.parametrize("n", [65, 66, 100, 128, 129, 200, 300, 1000, 2048])
def test_sy2sb_preserves_spectrum(n):
    from vivit_amd import kernels

    g = torch.Generator().manual_seed(n)
    M = torch.randn(n, n, generator=g, dtype=torch.float64)
    S = ((M + M.T) / 2).float()
    AB, tau1, A = kernels.sy2sb(S.to(DEV))
    ABh = AB.cpu().double().numpy()
    assert np.abs(ABh[:, :NB]).max() == 0.0  # bulge room is zero
    band = from_rowband(ABh)
    ref = np.linalg.eigvalsh(S.double().numpy())
    w = np.linalg.eigvalsh(band)
    scale = np.abs(ref).max()
    assert np.abs(w - ref).max() <= 5e-6 * scale, np.abs(w - ref).max() / scale
    if n <= 300:
        # S = Q1 B Q1^T with Q1 from the stored reflector rows
        Ah, t1 = A.cpu().double().numpy(), tau1.cpu().double().numpy()
        Q = np.eye(n)
        for j0 in range(0, n - NB, NB):
            mp = n - j0 - NB
            for c in range(min(NB, mp)):
                v = np.zeros(n)
                v[j0 + NB + c :] = Ah[j0 + c, j0 + NB + c :]
                assert abs(v[j0 + NB + c] - 1.0) < 1e-12
                Q = Q - t1[j0 + c] * np.outer(Q @ v, v)
        assert np.abs(Q @ band @ Q.T - S.double().numpy()).max() <= 3e-5 * scale


@pytest.mark.parametrize("n", [2048, 3000])
@pytest.mark.parametrize("kind", ["dense", "lowrank", "decay"])
def test_two_stage_eigenvalues(kind, n):
    """values-only symeig takes the two-stage path for n >= 2048."""
    from test_symeig_large_gpu import make_matrix
    from vivit_amd import kernels

    S = make_matrix(kind, n)
    ref = np.linalg.eigvalsh(S.double().numpy())
    w, _ = kernels.symeig(S.to(DEV), eigenvectors=False)
    assert np.abs(w.cpu().double().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("n", [193, 300, 1000, 2048])
@pytest.mark.parametrize("kind", ["dense", "lowrank", "decay", "clustered"])
def test_two_stage_eigenvectors(kind, n, monkeypatch):
    """Full two-stage path with vectors (band reduction, bulge chasing, D&C, Q2 and Q1 back-transforms)."""
    import subprocess, sys, os, json

    # the path is chosen once per process through VIVIT_TWO_STAGE: run in a child process
    code = f"""
import sys, json, numpy as np, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
from test_symeig_large_gpu import make_matrix
from vivit_amd import kernels
S = make_matrix({kind!r}, {n})
ref = np.linalg.eigvalsh(S.double().numpy())
w, Z = kernels.symeig(S.to("cuda:0"), eigenvectors=True)
w = w.cpu().double().numpy(); Zc = Z.cpu().double().numpy()
scale = np.abs(ref).max()
print(json.dumps(dict(eig=float(np.abs(w - ref).max() / scale), orth=float(np.abs(Zc.T @ Zc - np.eye({n})).max()),
      resid=float(np.abs(S.double().numpy() @ Zc - Zc * w[None, :]).max() / scale))))
"""
    env = dict(os.environ, VIVIT_TWO_STAGE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["eig"] <= 1e-5, res
    assert res["orth"] <= 5e-5, res
    assert res["resid"] <= 3e-5, res


@pytest.mark.parametrize("two_stage", ["0", "1"])
@pytest.mark.parametrize("n,world", [(1000, 3), (2048, 8), (100, 2)])
def test_symeig_rows_virtual_ranks(n, world, two_stage):
    """vivit_symeig_rows_f32 (the per-rank unit of the multi-GPU eigensolver): the row slices of `world`
    virtual ranks, concatenated, are the eigenvector matrix of the full solve (both reductions)."""
    import subprocess, sys, os, json

    code = f"""
import sys, json, numpy as np, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
from test_symeig_large_gpu import make_matrix
from vivit_amd import kernels
from vivit_amd.distributed import row_slices
S = make_matrix("decay", {n})
Sd = S.to("cuda:0")
w_full, Z_full = kernels.symeig(Sd, eigenvectors=True)
parts = []
for (lo, hi) in row_slices({n}, {world}):
    w, Zt = kernels.symeig_rows(Sd, lo, hi)
    assert torch.equal(w, w_full)
    assert Zt.shape == (hi - lo, {n})
    parts.append(Zt)
Zt = torch.cat(parts)
# bit-identical: every row goes through the same arithmetic whatever the slice it is processed in
same = bool(torch.equal(Zt.T, Z_full))
diff = float((Zt.T - Z_full).abs().max())
Zc = Zt.T.cpu().double().numpy(); w = w_full.cpu().double().numpy()
scale = np.abs(w).max()
print(json.dumps(dict(same=same, diff=diff, orth=float(np.abs(Zc.T @ Zc - np.eye({n})).max()),
      resid=float(np.abs(S.double().numpy() @ Zc - Zc * w[None, :]).max() / scale))))
"""
    env = dict(os.environ, VIVIT_TWO_STAGE=two_stage)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["diff"] <= 2e-5, res   # split-K choices may depend on the row count: allow rounding-level differences
    assert res["orth"] <= 5e-5, res
    assert res["resid"] <= 3e-5, res


def test_two_stage_default_size_degenerate_inputs():
    """n = 8192 takes the two-stage path by default: NaN input must raise like a failing Tensor.symeig
    (vivit/utils/eig.py:37-40) instead of hanging or faulting; the zero matrix and a multiple of the identity
    (every reflector degenerate, total deflation) must come back exact."""
    from vivit_amd import kernels

    n = 8192
    g = torch.Generator().manual_seed(0)
    M = torch.randn(n, n, generator=g)
    S = ((M + M.T) / 2).to(DEV)
    S[100, 200] = float("nan")
    S[200, 100] = float("nan")
    for vec in (False, True):
        with pytest.raises(RuntimeError):
            kernels.symeig(S, eigenvectors=vec)
    w, Z = kernels.symeig(torch.zeros(n, n, device=DEV), eigenvectors=True)
    assert float(w.abs().max()) == 0.0
    assert float((Z.T @ Z - torch.eye(n, device=DEV)).abs().max()) < 1e-6
    w, Z = kernels.symeig(3.0 * torch.eye(n, device=DEV), eigenvectors=True)
    assert float((w - 3.0).abs().max()) == 0.0
    assert float((Z.T @ Z - torch.eye(n, device=DEV)).abs().max()) < 1e-6


@pytest.mark.parametrize("mp", [64, 100, 1000, 5000, 20000])
def test_panel_qr_entry(mp):
    """vivit_sy2sb_panel_qr_f32 on its own (the replicated step of the sharded band reduction): Q = I - V T V^T is
    orthogonal, Q^T A = [R; 0] with R = strict upper triangle left in the panel + betas, and |R| equals LAPACK's."""
    from vivit_amd import kernels

    g = torch.Generator().manual_seed(mp)
    A0 = torch.randn(mp, NB, generator=g)
    pan = A0.clone().to(DEV)
    Vt, tau, betas, T = kernels.panel_qr_(pan)
    V = Vt.double().cpu().T                      # [mp, 64]
    Td, A0d = T.double().cpu(), A0.double()
    ncol = min(mp, NB)
    R = torch.triu(pan[:ncol].double().cpu(), 1) + torch.diag(betas.double().cpu())[:ncol]
    # unit lower trapezoidal V
    assert torch.equal(torch.diagonal(V[:ncol]), torch.ones(ncol, dtype=torch.float64))
    assert float(torch.triu(V[:ncol], 1).abs().max()) == 0.0
    QtA = A0d - V @ (Td.T @ (V.T @ A0d))         # Q^T A = (I - V T^T V^T) A
    scale = float(A0d.abs().max()) * (mp ** 0.5)
    assert float((QtA[:ncol] - R).abs().max()) <= 2e-6 * scale
    if mp > ncol:
        assert float(QtA[ncol:].abs().max()) <= 2e-6 * scale
    # T^-1 + T^-T = V^T V  (the defining identity of the compact-WY factor of an orthogonal Q)
    S = V.T @ V
    kk = int((tau != 0).sum())                   # (a square panel's last reflector is the identity: tau = 0)
    assert kk >= ncol - 1 and bool((tau[:kk] != 0).all())
    Tinv = torch.linalg.inv(Td[:kk, :kk])
    assert float((Tinv + Tinv.T - S[:kk, :kk]).abs().max()) <= 1e-4
    Rl = np.linalg.qr(A0d.numpy(), mode="r")
    assert np.abs(np.abs(Rl[:ncol]) - np.abs(R.numpy())).max() <= 2e-6 * scale
