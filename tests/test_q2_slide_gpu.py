"""The sliding-window Q2 back-transformation on the bf16 pipe (csrc/q2slide.hip, vivit_q2_apply_f32 mode 1) against the
sequential application of every bulge-chasing reflector in fp64 (small n), against the block-step fp32 kernels
(mode 0) and through its size-independent property (orthonormal rows stay orthonormal).  Replaces the eigenvector half
of Tensor.symeig, vivit/linalg/eigh.py:248-250."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
DEV = "cuda:0"
NB = 64


def to_rowband(band):
    n = band.shape[0]
    AB = np.zeros((n, 2 * NB + 1), np.float32)
    for i in range(n):
        lo = max(0, i - NB)
        AB[i, lo - i + 2 * NB: 2 * NB + 1] = band[i, lo: i + 1]
    return AB


def chase(n, seed):
    from vivit_amd import kernels

    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    M = (M + M.T) / 2
    band = np.triu(np.tril(M, NB), -NB).astype(np.float32)
    d, e, R2, tau2 = kernels.sb2st(torch.from_numpy(to_rowband(band)).to(DEV))
    return band, d, e, R2, tau2


def apply_reference(Zt, R2, tau2):
    """Zt <- Zt Q2^T by the sequential reflectors in fp64: last sweep first, within a sweep group of 64 ... any valid
    order gives the same product; this one is the reverse generation order."""
    n = R2.shape[0]
    Z = Zt.astype(np.float64).copy()
    for s in range(n - 3, -1, -1):
        kmax = 0
        while s + 1 + (kmax + 1) * NB < n:
            kmax += 1
        for k in range(kmax, -1, -1):
            c0 = s + 1 + k * NB
            L = min(NB, n - c0)
            v = R2[s, c0:c0 + L].astype(np.float64)
            Z[:, c0:c0 + L] -= tau2[s, k] * np.outer(Z[:, c0:c0 + L] @ v, v)
    return Z


@pytest.mark.parametrize("n,nrows", [(196, 16), (200, 37), (256, 5), (324, 48), (448, 64), (516, 130)])
def test_slide_matches_sequential_reflectors(n, nrows):
    from vivit_amd import kernels

    band, d, e, R2, tau2 = chase(n, n)
    rng = np.random.default_rng(n + 1)
    Z0 = (rng.standard_normal((nrows, n)) / np.sqrt(n)).astype(np.float32)
    ref = apply_reference(Z0, R2.cpu().numpy(), tau2.cpu().numpy())
    got = kernels.q2_apply_(torch.from_numpy(Z0).to(DEV).clone(), R2, tau2, mode=1).cpu().double().numpy()
    old = kernels.q2_apply_(torch.from_numpy(Z0).to(DEV).clone(), R2, tau2, mode=0).cpu().double().numpy()
    scale = np.abs(ref).max()
    err_new, err_old = np.abs(got - ref).max() / scale, np.abs(old - ref).max() / scale
    assert err_old <= 3e-6, err_old
    assert err_new <= 3e-6, (err_new, err_old)
    assert err_new <= 3.0 * err_old + 2e-7, (err_new, err_old)   # fp32-class arithmetic: not worse than the fp32 MFMA kernel


@pytest.mark.parametrize("n,nrows", [(1000, 1000), (1024, 300), (2052, 2052), (3000, 8192), (1216, 40960)])
def test_slide_matches_block_steps_and_keeps_rows_orthonormal(n, nrows):
    """mode 1 against mode 0 on the same reflectors; orthonormal rows (a slice of an orthogonal matrix, repeated with
    sign flips when more rows than columns are wanted: several slabs, 1 to 10 waves per workgroup)."""
    from vivit_amd import kernels

    band, d, e, R2, tau2 = chase(n, n)
    g = torch.Generator().manual_seed(n)
    Q = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))[0].float()
    reps = -(-nrows // n)
    Z0 = torch.cat([Q * (1.0 if r % 2 == 0 else -1.0) for r in range(reps)], 0)[:nrows].contiguous().to(DEV)
    new = kernels.q2_apply_(Z0.clone(), R2, tau2, mode=1)
    old = kernels.q2_apply_(Z0.clone(), R2, tau2, mode=0)
    diff = (new - old).abs().max().item() / old.abs().max().item()
    assert diff <= 5e-6, diff
    m = min(nrows, n)
    gram = new[:m].double() @ new[:m].double().T
    orth = (gram - torch.eye(m, dtype=torch.float64, device=DEV)).abs().max().item()
    gram_old = old[:m].double() @ old[:m].double().T
    orth_old = (gram_old - torch.eye(m, dtype=torch.float64, device=DEV)).abs().max().item()
    assert orth <= 2e-5, (orth, orth_old)
    assert orth <= 3.0 * orth_old + 1e-6, (orth, orth_old)
    # the band matrix is reproduced: rows of (Q_T^T Q2^T) are its eigenvectors when Q_T are those of the tridiagonal
    if nrows == n and n <= 2052:
        T = np.diag(d.cpu().double().numpy()) + np.diag(e.cpu().double().numpy(), 1) + np.diag(e.cpu().double().numpy(), -1)
        w, V = np.linalg.eigh(T)
        Zt = kernels.q2_apply_(torch.from_numpy(V.T.astype(np.float32)).contiguous().to(DEV), R2, tau2, mode=1).cpu().double().numpy()
        resid = np.abs(Zt @ band.astype(np.float64) - w[:, None] * Zt).max() / np.abs(w).max()
        assert resid <= 2e-5, resid


def test_slide_is_bit_reproducible():
    from vivit_amd import kernels

    band, d, e, R2, tau2 = chase(1000, 3)
    g = torch.Generator().manual_seed(5)
    Z0 = torch.randn(4096, 1000, generator=g).to(DEV)
    a = kernels.q2_apply_(Z0.clone(), R2, tau2, mode=1)
    b = kernels.q2_apply_(Z0.clone(), R2, tau2, mode=1)
    assert torch.equal(a, b)


def test_slide_refuses_unaligned_shapes():
    from vivit_amd import _lib, kernels

    band, d, e, R2, tau2 = chase(198, 1)     # n % 4 != 0
    Z0 = torch.randn(16, 198, device=DEV)
    with pytest.raises(_lib.VivitHipError) as exc:
        kernels.q2_apply_(Z0.clone(), R2, tau2, mode=1)
    assert exc.value.status == _lib.VIVIT_E_UNSUPPORTED
    out = kernels.q2_apply_(Z0.clone(), R2, tau2, mode=-1)   # the solver's choice falls back to the block steps
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("waves,loaders,lockstep", [(5, 0, None), (11, 0, None), (12, 0, None), (5, 2, None), (10, 2, None), (10, 2, 0), (5, 2, 1), (8, 1, 64)])
def test_slide_wave_configurations(tmp_path, waves, loaders, lockstep):
    """the self-load path of qs_apply_kernel (no loader waves: LOADERS=0, or more than ten compute waves --
    which the product only takes for > 160 rows per CU, i.e. nrows > 40 960 or a CU-masked partition) has hand-counted
    s_waitcnt vmcnt(4) waits that no test reached.  Child processes (the knobs are read once per process) force 5 / 11 / 12
    compute waves without loaders, and two loader configurations for comparison: against the fp64 sequential reflectors,
    against the block-step kernels, orthonormality, bit-reproducibility.  Round 6: the loaders of an XCD keep in lock-step through
    a progress board (VIVIT_Q2_LOCKSTEP = window in blocks, default 4): off, the tightest window and a loose one must give the
    same results -- the throttle only delays image requests."""
    import json
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    out = tmp_path / "q2.json"
    subprocess.run([sys.executable, os.path.join(here, "q2_slide_child.py"), str(out)], check=True, timeout=600,
                   env=dict(os.environ, VIVIT_Q2_SLIDE_WAVES=str(waves), VIVIT_Q2_SLIDE_LOADERS=str(loaders),
                            **({} if lockstep is None else {"VIVIT_Q2_LOCKSTEP": str(lockstep)})))
    res = json.loads(out.read_text())
    for key, row in res["sequential"].items():
        assert row["block"] <= 3e-6 and row["slide"] <= 3e-6, (key, row)
        assert row["slide"] <= 3.0 * row["block"] + 2e-7, (key, row)
    for key, row in res["block_steps"].items():
        assert row["diff"] <= 5e-6 and row["orth"] <= 5e-6 and row["bitwise_repeat"], (key, row)
