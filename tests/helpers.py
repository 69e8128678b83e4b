"""Shared test helpers: golden-vector loading, fake BackPACK modules, an oracle-backed kernel
backend for host-logic tests on CPU (TEST INFRASTRUCTURE -- the product never imports this)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["mlp_small", "conv_like", "subsampled", "mc1", "wide", "multikernel"]


def load_golden(name):
    with np.load(os.path.join(GOLDEN, f"{name}.npz")) as z:
        return {k: z[k] for k in z.files}


BIG_CASES = ["two_stage"]   # factors regenerated from their seed (tests/golden/make_golden.py:BIG_CASES)


def planted_factors(seed, C, N, shapes, planted=10):
    """Seeded factors of the BIG cases (shared by tests/golden/make_golden.py, which records the reference's outputs on them,
    and by the tests, which regenerate them): ``randn / sqrt(N)`` like the small cases, plus ``planted`` rank-one terms in the
    first parameter whose Gram eigenvalues 100 * 1.5^j stand clear of the bulk edge (~ 43) and of each other -- with plain
    noise the top eigenvalues crowd at the Marchenko-Pastur edge and fp32 eigenvectors (the reference's own LAPACK ones
    included) are only defined to ~ eps lambda_max / gap, which would turn the vector-valued comparisons into noise."""
    g = torch.Generator().manual_seed(seed)
    V = [torch.randn(C, N, *s, generator=g, dtype=torch.float32) / (N ** 0.5) for s in shapes]
    G = [torch.randn(N, *s, generator=g, dtype=torch.float32) / N for s in shapes]
    for j in range(planted):
        u = torch.randn(C, N, generator=g, dtype=torch.float32)
        w = torch.randn(*shapes[0], generator=g, dtype=torch.float32)
        amp = (100.0 * 1.5 ** j) ** 0.5
        V[0] += amp * (u / u.norm()).reshape(C, N, *([1] * len(shapes[0]))) * (w / w.norm())
    return V, G


def seeded_factors(gold, device="cpu"):
    """The factors of a fixture that stores only their seed (torch's CPU generator; the build container and the GPU box run
    the same torch), checked against the stored fingerprints."""
    flat = [int(x) for x in gold["shapes_flat"]]
    shapes, i = [], 0
    while i < len(flat):
        shapes.append(tuple(flat[i + 1:i + 1 + flat[i]]))
        i += 1 + flat[i]
    V, G = planted_factors(int(gold["seed"]), int(gold["C"]), int(gold["N"]), shapes)
    have = [float(v.double().sum()) for v in V] + [float(v.double().abs().sum()) for v in V]
    np.testing.assert_allclose(have, gold["V_checksum"], rtol=1e-9, err_msg="this torch's CPU generator differs from the one that recorded the fixture")
    have = [float(x.double().sum()) for x in G] + [float(x.double().abs().sum()) for x in G]
    np.testing.assert_allclose(have, gold["g_checksum"], rtol=1e-9)
    return [v.to(device) for v in V], [x.to(device) for x in G]


def golden_factors(gold, device="cpu"):
    if "seed" in gold and "V0" not in gold:
        return seeded_factors(gold, device)
    V, G = [], []
    i = 0
    while f"V{i}" in gold:
        V.append(torch.from_numpy(gold[f"V{i}"]).to(device))
        G.append(torch.from_numpy(gold[f"g{i}"]).to(device))
        i += 1
    return V, G


class FakeModule(torch.nn.Module):
    """Leaf module holding parameters + the ``input0`` BackPACK stores (vivit/linalg/utils.py:54)."""

    def __init__(self, params, batch_size):
        super().__init__()
        for i, p in enumerate(params):
            self.register_parameter(f"p{i}", p)
        self.input0 = torch.zeros(batch_size, 1)


def top_k_criterion(k, must_exceed=1e-5):
    """Rule of the reference tests' make_criterion (test/optim/settings.py:21-47)."""

    def criterion(evals):
        n = len(evals)
        shift = max(n - k, 0)
        return [i + shift for i, ev in enumerate(evals[shift:]) if ev > must_exceed]

    return criterion


def keep_all(evals):
    return list(range(evals.numel()))


def constant_damping(d):
    def damping(evals, evecs, gammas, lambdas):
        return d * torch.ones_like(evals)

    return damping


class OracleBackend:
    """Same function names as ``vivit_amd.kernels``, computed with torch on CPU (oracle ops).
    ``set_kernel_backend`` monkeypatches it over the launchers to test the Python hook layer without a GPU."""

    @staticmethod
    def _acc(res, out, alpha, beta):
        res = alpha * res
        if out is None:
            return res
        out.copy_(res + beta * out if beta != 0.0 else res)
        return out

    def gram_syrk(self, A, out=None, alpha=1.0, beta=0.0):
        return self._acc(A @ A.T, out, alpha, beta)

    def gemm_nt(self, A, B, out=None, alpha=1.0, beta=0.0):
        return self._acc(A @ B.T, out, alpha, beta)

    def gemm_nn(self, A, B, out=None, alpha=1.0, beta=0.0):
        return self._acc(A @ B, out, alpha, beta)

    def gemm_tn(self, A, B, out=None, alpha=1.0, beta=0.0):
        return self._acc(A.T @ B, out, alpha, beta)

    def gram_hadamard(self, Gz, Gs, C, N, out=None, alpha=1.0, beta=0.0):
        res = (Gz.view(1, N, 1, N) * Gs.view(C, N, C, N)).reshape(C * N, C * N)
        return self._acc(res, out, alpha, beta)

    def gram_hadamard_block(self, Gz, Gs, Cr, Nr, Cc, Nc, out=None, alpha=1.0, beta=0.0):
        res = (Gz.view(1, Nr, 1, Nc) * Gs.reshape(Cr, Nr, Cc, Nc)).reshape(Cr * Nr, Cc * Nc)
        return self._acc(res, out, alpha, beta)

    def class_contract(self, mat, s):
        return torch.einsum("vcn,cno->von", mat, s).contiguous()

    def class_expand(self, s, U):
        return torch.einsum("cno,von->vcn", s, U).contiguous()

    def linear_weight_mjp(self, s, z):
        return torch.einsum("vno,ni->vnoi", s, z)

    def conv2d_weight_mjp(self, M, x, kernel_size, stride, padding, dilation):
        xu = torch.nn.functional.unfold(x, kernel_size, dilation=dilation, padding=padding, stride=stride)
        out = torch.einsum("vnol,nkl->vnok", M.flatten(3), xu)
        return out.reshape(*M.shape[:3], x.shape[1], *kernel_size)

    def symeig(self, G, eigenvectors=False, overwrite=False):
        from oracle import vivit_oracle as oracle

        w, Z = oracle.tensor_symeig(G, eigenvectors=eigenvectors, upper=False)
        return w, (Z if eigenvectors else None)

    def symeig_reduce(self, G, overwrite=False):
        from vivit_amd.kernels import SymeigPlan

        w, Z = self.symeig(G, True)
        return SymeigPlan(w, G.shape[0], full=Z)

    def symeig_rows(self, G, row_begin, row_end, overwrite=False):
        w, Z = self.symeig(G, True)
        return w, Z.T[row_begin:row_end].contiguous()

    # ---- the pieces of the sharded band reduction (vivit_amd/distributed.py:sy2sb_sharded_), LAPACK conventions in fp64
    def symeig_prepare_(self, A):
        A.copy_(torch.tril(A) + torch.tril(A, -1).T)
        scal = torch.zeros(16)
        scal[1] = 1.0
        return scal

    def panel_qr_(self, pan):
        NB = 64
        mp = pan.shape[0]
        x = pan.double()
        Vt = torch.zeros(NB, mp, dtype=torch.float64)
        tau = torch.zeros(NB, dtype=torch.float64)
        betas = torch.zeros(NB, dtype=torch.float64)
        for c in range(min(mp, NB)):
            alpha = x[c, c].item()
            ssq = float((x[c + 1:, c] ** 2).sum())
            beta, t, sc = alpha, 0.0, 0.0
            if ssq > 0.0:
                beta = -float(np.copysign(np.sqrt(alpha * alpha + ssq), alpha))
                t = (beta - alpha) / beta
                sc = 1.0 / (alpha - beta)
            v = torch.zeros(mp, dtype=torch.float64)
            v[c] = 1.0
            v[c + 1:] = x[c + 1:, c] * sc
            z = t * (v @ x)
            z[: c + 1] = 0.0
            x -= torch.outer(v, z)
            Vt[c], tau[c], betas[c] = v, t, beta
        T = torch.zeros(NB, NB, dtype=torch.float64)
        for j in range(NB):   # larft, forward / columnwise: Q = H_0 H_1 ... = I - V T V^T
            T[j, j] = tau[j]
            if j:
                T[:j, j] = -tau[j] * (T[:j, :j] @ (Vt[:j] @ Vt[j]))
        pan.copy_(x.float())
        return Vt.float(), tau.float(), betas.float(), T.float()

    def symeig_banded_rows(self, A, tau1, scal, row_begin, row_end):
        NB = 64
        n = A.shape[0]
        Ad = A.double()
        B = torch.zeros(n, n, dtype=torch.float64)
        for i in range(n):
            lo = max(0, i - NB)
            B[i, lo: i + 1] = Ad[i, lo: i + 1]
        B = B + torch.tril(B, -1).T
        w, Z = torch.linalg.eigh(B)
        # Q1 = product of the panel reflectors in generation order; eigenvectors of the full matrix = Q1 Z
        refl = []
        for j0 in range(0, n - NB, NB):
            gi = j0 + NB
            for c in range(min(NB, n - gi)):
                v = torch.zeros(n, dtype=torch.float64)
                v[gi + c] = 1.0
                v[gi + c + 1:] = Ad[j0 + c, gi + c + 1:]
                refl.append((v, float(tau1[j0 + c])))
        for v, t in reversed(refl):
            Z -= t * torch.outer(v, v @ Z)
        return (w / float(scal[1])).float(), Z.T[row_begin:row_end].float().contiguous()

    def pack_lower(self, G):
        i, j = torch.tril_indices(G.shape[0], G.shape[0])
        return G[i, j].contiguous()

    def unpack_lower_(self, packed, G):
        i, j = torch.tril_indices(G.shape[0], G.shape[0])
        G[i, j] = packed
        G[j, i] = packed
        return G

    def dir_curvature(self, GE, evals, C, N, scale):
        K = evals.numel()
        return scale * (GE.view(C, N, K) ** 2).sum(0) / evals

    def scale_cols_rsqrt_(self, X, evals, pre=1.0):
        X.mul_(pre / evals.sqrt())
        return X

    def normalize_rows_(self, tensors):
        K = tensors[0].shape[0]
        sq = sum((t.reshape(K, -1) ** 2).sum(1) for t in tensors)
        for t in tensors:
            t.mul_((1 / sq.sqrt()).view(K, *([1] * (t.dim() - 1))))
        return tensors


_PATCHED = {}


def set_kernel_backend(backend):
    """Monkeypatch the launcher functions of ``vivit_amd.kernels`` with ``backend``'s methods of the same names
    (``None`` restores the HIP launchers).  The seam lives here, in tests/: the product module has no switch."""
    from vivit_amd import kernels

    for name, fn in _PATCHED.items():
        setattr(kernels, name, fn)
    _PATCHED.clear()
    if backend is None:
        return
    for name in dir(backend):
        if name.startswith("_") or not callable(getattr(backend, name)):
            continue
        _PATCHED[name] = getattr(kernels, name)
        setattr(kernels, name, getattr(backend, name))


def resnet32(num_classes=100):
    """CIFAR ResNet-32 of BASELINE config 4 (defined next to the benchmark that times it)."""
    import bench_configs

    return bench_configs.resnet32(num_classes)
