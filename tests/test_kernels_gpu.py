"""GPU parity of the raw kernels (through the C ABI) against fp64 references computed on the host."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _rel(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


@pytest.mark.parametrize(
    "m,n,k",
    [(1, 1, 1), (5, 7, 3), (128, 128, 16), (130, 250, 37), (64, 300, 1000), (257, 129, 515), (10, 10, 5000), (300, 300, 70000),
     (64, 1024, 4096), (33, 515, 129), (64, 2051, 20000), (17, 260, 64),
     # 64-row streaming kernel (gemm64_dma_kernel): N, K >= 2048, K % 16 == 0; ragged N, M < 64, deep K (splits)
     (64, 2048, 2048), (36, 2100, 6400), (4, 4096, 40000), (64, 2304, 33 * 1024),
     # its bf16-pipe form (gemm64_bx_kernel): an odd number of K tiles per split, a chain boundary (2048 k) plus one tile
     (64, 2048, 2064), (60, 2560, 3 * 2048 + 16)],
)
def test_gemm_variants(m, n, k):
    from vivit_amd import kernels

    g = torch.Generator().manual_seed(m * 1000003 + n * 1009 + k)
    A = torch.randn(m, k, generator=g)
    B = torch.randn(n, k, generator=g)
    C0 = torch.randn(m, n, generator=g)
    ref = A.double() @ B.double().T
    Ad, Bd = A.to(_dev()), B.to(_dev())
    tol = 2e-6 * (k ** 0.5) + 1e-6
    out = kernels.gemm_nt(Ad, Bd)
    assert _rel(out, ref) < tol
    out = kernels.gemm_nn(Ad, Bd.T.contiguous())
    assert _rel(out, ref) < tol
    out = kernels.gemm_tn(Ad.T.contiguous(), Bd.T.contiguous())
    assert _rel(out, ref) < tol
    # alpha / beta
    Cd = C0.to(_dev()).clone()
    kernels.gemm_nt(Ad, Bd, out=Cd, alpha=0.5, beta=-2.0)
    assert _rel(Cd, 0.5 * ref - 2.0 * C0.double()) < tol


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C write (MFMA C/D layout)."""
    from vivit_amd import kernels

    n = 96
    B = torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251
    I = torch.eye(n)
    out = kernels.gemm_nn(I.to(_dev()), B.to(_dev()))
    assert torch.equal(out.cpu(), B)
    out = kernels.gemm_nt(I.to(_dev()), B.to(_dev()))
    assert torch.equal(out.cpu(), B.T)


@pytest.mark.parametrize(
    "m,n,k",
    # outputs with >= 200 tiles of 256 x 256: the large-tile kernel (gemm256_kernel), full and ragged
    # edge tiles, K with and without a tail (< 16) and longer than one flush period (2048)
    [(4096, 4096, 1024), (3600, 3604, 1043), (3840, 3584, 4200), (3588, 4000, 8192 + 1024 + 16)],
)
def test_gemm_large_tile(m, n, k):
    from vivit_amd import kernels

    g = torch.Generator().manual_seed(m + 3 * n + 7 * k)
    A = torch.randn(m, k, generator=g)
    B = torch.randn(n, k, generator=g)
    C0 = torch.randn(m, n, generator=g)
    Ad, Bd = A.to(_dev()), B.to(_dev())
    ref = (Ad.double() @ Bd.double().T).cpu()
    tol = 2e-6 * (k ** 0.5) + 1e-6
    assert _rel(kernels.gemm_nt(Ad, Bd), ref) < tol
    assert _rel(kernels.gemm_nn(Ad, Bd.T.contiguous()), ref) < tol
    assert _rel(kernels.gemm_tn(Ad.T.contiguous(), Bd.T.contiguous()), ref) < tol
    Cd = C0.to(_dev()).clone()
    kernels.gemm_nt(Ad, Bd, out=Cd, alpha=0.5, beta=-2.0)
    assert _rel(Cd, 0.5 * ref - 2.0 * C0.double()) < tol


@pytest.mark.parametrize("n,p", [(5120, 1040), (5000, 9117), (1280, 40000), (1024, 70016), (2000, 12345)])  # last three: split-K slabs
def test_gram_syrk_large_tile(n, p):
    from vivit_amd import kernels

    g = torch.Generator().manual_seed(n + p)
    Ad = torch.randn(n, p, generator=g).to(_dev())
    ref = (Ad.double() @ Ad.double().T).cpu()
    G = kernels.gram_syrk(Ad)
    tol = 2e-6 * (p ** 0.5) + 1e-6
    assert _rel(G, ref) < tol
    assert torch.equal(G, G.T), "Gram must be exactly symmetric"
    G2 = kernels.gram_syrk(Ad, out=G.clone(), alpha=1.0, beta=1.0)
    assert _rel(G2, 2 * ref) < tol
    assert torch.equal(G2, G2.T)


@pytest.mark.parametrize("n,p", [(1, 1), (3, 10), (15, 42), (128, 64), (200, 333), (640, 513), (1280, 4096), (300, 100000)])
def test_gram_syrk(n, p):
    from vivit_amd import kernels

    g = torch.Generator().manual_seed(n * 7919 + p)
    A = torch.randn(n, p, generator=g)
    ref = A.double() @ A.double().T
    Ad = A.to(_dev())
    G = kernels.gram_syrk(Ad)
    tol = 2e-6 * (p ** 0.5) + 1e-6
    assert _rel(G, ref) < tol
    assert torch.equal(G, G.T), "Gram must be exactly symmetric"
    # accumulate (beta = 1), the `gram += gram_p` of vivit/utils/gram.py:104-116
    G2 = kernels.gram_syrk(Ad, out=G.clone(), alpha=1.0, beta=1.0)
    assert _rel(G2, 2 * ref) < tol
    assert torch.equal(G2, G2.T)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 15, 33, 64, 100, 128, 192])
@pytest.mark.parametrize("kind", ["dense", "lowrank"])
def test_symeig_small(n, kind):
    from vivit_amd import kernels

    g = torch.Generator().manual_seed(n * 31 + (kind == "dense"))
    if kind == "dense":
        M = torch.randn(n, n, generator=g)
        S = (M + M.T) / 2
    else:
        r = max(1, n // 3)
        V = torch.randn(n, r, generator=g)
        S = V @ V.T
    ref_w, ref_Z = torch.linalg.eigh(S.double())
    Sd = S.to(_dev())
    w, _ = kernels.symeig(Sd, eigenvectors=False)
    scale = ref_w.abs().max().item()
    assert (w.double().cpu() - ref_w).abs().max().item() <= 2e-5 * scale + 1e-30
    assert torch.equal(Sd.cpu(), S), "input must not be modified"
    w2, Z = kernels.symeig(Sd, eigenvectors=True)
    assert (w2.double().cpu() - ref_w).abs().max().item() <= 2e-5 * scale + 1e-30
    Zc = Z.double().cpu()
    eye = torch.eye(n, dtype=torch.float64)
    assert (Zc.T @ Zc - eye).abs().max().item() < 2e-5
    resid = S.double() @ Zc - Zc * w2.double().cpu()
    assert resid.abs().max().item() <= 5e-5 * scale


def test_symeig_rank1_denormal():
    """Analogue of the reference's tensor_causes_symeig_error.pt (test/utils/test_stable_symeig.py:13-45):
    128x128, effectively rank one with lambda_max ~ 6e7 on a background of denormals."""
    from vivit_amd import kernels

    g = torch.Generator().manual_seed(0)
    u = torch.randn(128, generator=g)
    u = u / u.norm()
    S = 6.2764e7 * torch.outer(u, u)
    S = S + 1.8367e-40 * torch.ones(128, 128)
    S = (S + S.T) / 2
    w, Z = kernels.symeig(S.to(_dev()), eigenvectors=True)
    assert abs(w[-1].item() - 6.2764e7) < 6.2764e7 * 1e-5
    assert w[:-1].abs().max().item() < 6.2764e7 * 1e-5
    Zc = Z.double().cpu()
    assert (Zc.T @ Zc - torch.eye(128, dtype=torch.float64)).abs().max().item() < 2e-5


def test_symeig_nan_raises():
    from vivit_amd import kernels

    S = torch.eye(8)
    S[2, 1] = float("nan")
    with pytest.raises(RuntimeError):
        kernels.symeig(S.to(_dev()), eigenvectors=True)


def test_cpu_tensor_rejected():
    from vivit_amd import kernels

    with pytest.raises(RuntimeError):
        kernels.gram_syrk(torch.randn(4, 4))


@pytest.mark.parametrize("K,n,P", [(1, 40, 10), (1, 3000, 5000), (3, 1025, 1027), (8, 2048, 4096), (16, 777, 333), (10, 1280, 7840)])
def test_backproject_skinny(K, n, P):
    """K7/K8 with few directions (streaming kernel behind gemm_nn for m <= 16)."""
    from vivit_amd import kernels

    g = torch.Generator().manual_seed(K * 131 + n + P)
    E = torch.randn(K, n, generator=g)
    V = torch.randn(n, P, generator=g)
    C0 = torch.randn(K, P, generator=g)
    ref = E.double() @ V.double()
    out = kernels.gemm_nn(E.to(_dev()), V.to(_dev()))
    tol = 2e-6 * (n ** 0.5) + 1e-6
    assert _rel(out, ref) < tol
    Cd = C0.to(_dev()).clone()
    kernels.gemm_nn(E.to(_dev()), V.to(_dev()), out=Cd, alpha=-0.5, beta=2.0)
    assert _rel(Cd, -0.5 * ref + 2.0 * C0.double()) < tol
    out2 = kernels.gemm_nn(E.to(_dev()), V.to(_dev()))
    assert torch.equal(out, out2), "fixed-order slab reduction must be bit-reproducible"


@pytest.mark.parametrize("C,N,O,I", [(3, 5, 7, 9), (10, 64, 16, 32), (1, 33, 5, 12)])
def test_linear_weight_mjp(C, N, O, I):
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C * N)
    s, z = torch.randn(C, N, O, generator=g), torch.randn(N, I, generator=g)
    V = kernels.linear_weight_mjp(s.to(dev), z.to(dev))
    assert torch.equal(V.cpu(), torch.einsum("vno,ni->vnoi", s, z))       # one multiply per entry: bit-exact


@pytest.mark.parametrize("geom", [
    # (V, N, Cin, H, W, Cout, k, stride, padding, dilation)
    (2, 3, 3, 8, 8, 2, (2, 2), (1, 1), (0, 0), (1, 1)),
    (10, 4, 3, 32, 32, 6, (5, 5), (1, 1), (0, 0), (1, 1)),        # LeNet conv1: 784 output positions (> one LDS chunk)
    (1, 5, 16, 16, 16, 32, (3, 3), (2, 2), (1, 1), (1, 1)),       # ResNet stage transition
    (3, 2, 4, 9, 7, 5, (3, 2), (2, 1), (2, 0), (2, 3)),           # everything odd
])
def test_conv2d_weight_mjp(geom):
    from vivit_amd import kernels

    Vd, N, Cin, H, W, Cout, k, stride, padding, dilation = geom
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(geom[:6]))
    x = torch.randn(N, Cin, H, W, generator=g)
    OH = (H + 2 * padding[0] - dilation[0] * (k[0] - 1) - 1) // stride[0] + 1
    OW = (W + 2 * padding[1] - dilation[1] * (k[1] - 1) - 1) // stride[1] + 1
    M = torch.randn(Vd, N, Cout, OH, OW, generator=g)
    out = kernels.conv2d_weight_mjp(M.to(dev), x.to(dev), k, stride, padding, dilation)
    xu = torch.nn.functional.unfold(x.double(), k, dilation=dilation, padding=padding, stride=stride)
    ref = torch.einsum("vnol,nkl->vnok", M.double().flatten(3), xu).reshape(Vd, N, Cout, Cin, *k)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * ref.abs().max().item())


@pytest.mark.parametrize("kind", ["nt", "nn", "tn", "syrk"])
def test_large_products_on_the_256_tile_path(kind):
    """Products big enough for the 256 x 256 tile kernels (>= 200 tiles, K >= 1024), i.e. the bf16-pipe path with the
    exact three-way operand split (default) for every operand layout, ragged edges included; sampled against fp64."""
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    m, n, k = 4100, 3900, 2064
    if kind == "nt":
        A, B = torch.randn(m, k, device=dev, generator=g), torch.randn(n, k, device=dev, generator=g)
        C = kernels.gemm_nt(A, B)
        ref = lambda I, J: A[I].double() @ B[J].double().T  # noqa: E731
    elif kind == "nn":
        A, B = torch.randn(m, k, device=dev, generator=g), torch.randn(k, n, device=dev, generator=g)
        C = kernels.gemm_nn(A, B)
        ref = lambda I, J: A[I].double() @ B[:, J].double()  # noqa: E731
    elif kind == "tn":
        A, B = torch.randn(k, m, device=dev, generator=g), torch.randn(k, n, device=dev, generator=g)
        C = kernels.gemm_tn(A, B)
        ref = lambda I, J: A[:, I].double().T @ B[:, J].double()  # noqa: E731
    else:
        m = n = 4100
        A = torch.randn(m, k, device=dev, generator=g)
        C0 = torch.randn(m, m, device=dev, generator=g)
        C0 = C0 + C0.T
        C = kernels.gram_syrk(A, out=C0.clone(), alpha=0.5, beta=2.0)
        assert torch.equal(C, C.T)
        ref = lambda I, J: 0.5 * (A[I].double() @ A[J].double().T) + 2.0 * C0[I][:, J].double()  # noqa: E731
    I = torch.tensor([0, 1, 255, 256, 257, 2047, 2048, m - 257, m - 2, m - 1], device=dev)
    J = torch.arange(0, n, 13, device=dev)
    err = (C[I][:, J].double() - ref(I, J)).abs().max().item()
    assert err <= 2e-5 * k ** 0.5, err   # fp32 accumulation of k unit-variance products


@pytest.mark.parametrize("m,n,k,pad", [(64, 64, 40960, 0), (64, 384, 20480, 64), (33, 100, 4100, 0), (64, 1024, 2048, 0),
                                       (7, 64, 8196, 4), (64, 128, 2052, 0)])
def test_deep_k_small_output_product(m, n, k, pad):
    """``A B^T`` with at most 64 output rows and a deep K-contiguous contraction (the band reduction's panel Gram blocks:
    the dedicated split-K kernel with 128-k steps); ragged M / N / K tails, padded leading dimensions, alpha / beta."""
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(m + n + k)
    A = torch.randn(m, k + pad, device=dev, generator=g)[:, :k]
    B = torch.randn(n, k + pad, device=dev, generator=g)[:, :k]
    ref = A.double() @ B.double().T
    out = kernels.gemm_nt(A, B)
    scale = (A.double().norm(dim=1)[:, None] * B.double().norm(dim=1)[None, :])
    assert ((out.double() - ref).abs() / scale).max().item() < 2e-6
    C0 = torch.randn(m, n, device=dev, generator=g)
    out2 = kernels.gemm_nt(A, B, out=C0.clone(), alpha=-0.5, beta=2.0)
    assert ((out2.double() - (-0.5 * ref + 2.0 * C0.double())).abs() / (scale + 1.0)).max().item() < 2e-6


@pytest.mark.parametrize("kind,m,n,k", [("syrk", 1280, 1280, 40000), ("syrk", 300, 300, 100000), ("nt", 1000, 456, 20480),
                                        ("tn", 2416, 2416, 20480), ("tn", 456, 850, 16400), ("nn", 700, 1300, 32768),
                                        ("syrk", 1023, 1023, 65552)])
def test_small_output_deep_contraction_split_k(kind, m, n, k):
    """Small outputs (< 100 tiles of 256 x 256) with K >= 16384: the bf16-pipe split-K path (whole-operand pieces, k tiles
    divided over blockIdx.y, slab + fixed-order reduce with SYRK mirror), all operand layouts, ragged M / N / K, alpha / beta."""
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(m + 3 * n + k)
    C0 = torch.randn(m, n, device=dev, generator=g)
    if kind == "syrk":
        A = torch.randn(m, k, device=dev, generator=g)
        C0 = C0 + C0.T
        C = kernels.gram_syrk(A, out=C0.clone(), alpha=0.5, beta=-1.0)
        assert torch.equal(C, C.T)
        ref = 0.5 * (A.double() @ A.double().T) - C0.double()
        scale = A.double().norm(dim=1)[:, None] * A.double().norm(dim=1)[None, :]
    elif kind == "nt":
        A, B = torch.randn(m, k, device=dev, generator=g), torch.randn(n, k, device=dev, generator=g)
        C = kernels.gemm_nt(A, B, out=C0.clone(), alpha=0.5, beta=-1.0)
        ref = 0.5 * (A.double() @ B.double().T) - C0.double()
        scale = A.double().norm(dim=1)[:, None] * B.double().norm(dim=1)[None, :]
    elif kind == "tn":
        A, B = torch.randn(k, m, device=dev, generator=g), torch.randn(k, n, device=dev, generator=g)
        C = kernels.gemm_tn(A, B, out=C0.clone(), alpha=0.5, beta=-1.0)
        ref = 0.5 * (A.double().T @ B.double()) - C0.double()
        scale = A.double().norm(dim=0)[:, None] * B.double().norm(dim=0)[None, :]
    else:
        A, B = torch.randn(m, k, device=dev, generator=g), torch.randn(k, n, device=dev, generator=g)
        C = kernels.gemm_nn(A, B, out=C0.clone(), alpha=0.5, beta=-1.0)
        ref = 0.5 * (A.double() @ B.double()) - C0.double()
        scale = A.double().norm(dim=1)[:, None] * B.double().norm(dim=0)[None, :]
    assert ((C.double() - ref).abs() / (scale + 1.0)).max().item() < 5e-6
