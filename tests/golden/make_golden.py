"""Generate golden vectors by running the REFERENCE (f-dangel/vivit @ /root/reference) itself.

Runs only in the build container (the reference does not exist on the GPU box); the resulting
``*.npz`` files hold data only -- seeded inputs and the reference's outputs -- and are committed.

How the reference is made importable:
  * ``Tensor.symeig`` was removed from torch: it is shimmed with ``torch.linalg.eigh/eigvalsh``
    (same semantics: ascending eigenvalues, column eigenvectors, ``upper=True`` default);
  * ``backpack-for-pytorch`` is not installed: an in-memory stub package provides exactly the
    names ``vivit`` imports (savefields ``grad_batch`` / ``sqrt_ggn_exact`` / ``sqrt_ggn_mc`` /
    ``vivit_ggn_exact``).  BackPACK's own factor materialisation is NOT emulated: the factors
    ``V_t`` / ``g`` (and the ViViT closures, built from the reference's own pairwise_dot / Vmp /
    mVp) are hand-made seeded tensors attached to parameters, and the reference's hooks are
    driven exactly as BackPACK would drive them (``hook(module)`` per module).

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_symeig_shim():
    def symeig(self, eigenvectors=False, upper=True):
        uplo = "U" if upper else "L"
        if eigenvectors:
            w, v = torch.linalg.eigh(self, UPLO=uplo)
            return w, v
        return torch.linalg.eigvalsh(self, UPLO=uplo), self.new_empty(0)

    torch.Tensor.symeig = symeig


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _install_backpack_stub():
    bp = _mod("backpack")
    ext = _mod("backpack.extensions")
    bpe = _mod("backpack.extensions.backprop_extension")

    class BackpropExtension:
        def __init__(self, savefield=None, subsampling=None, **kw):
            self.savefield = savefield
            self._subsampling = subsampling

        def get_subsampling(self):
            return self._subsampling

    bpe.BackpropExtension = BackpropExtension

    def _simple(name, savefield):
        def __init__(self, subsampling=None, mc_samples=1):
            BackpropExtension.__init__(self, savefield=savefield, subsampling=subsampling)
            self._mc_samples = mc_samples

        return type(name, (BackpropExtension,), {"__init__": __init__})

    ext.BatchGrad = _simple("BatchGrad", "grad_batch")
    ext.SqrtGGNExact = _simple("SqrtGGNExact", "sqrt_ggn_exact")
    ext.SqrtGGNMC = _simple("SqrtGGNMC", "sqrt_ggn_mc")

    so = _mod("backpack.extensions.secondorder")
    base = _mod("backpack.extensions.secondorder.base")

    class SecondOrderBackpropExtension(BackpropExtension):
        def __init__(self, savefield, fail_mode, module_exts, subsampling=None):
            BackpropExtension.__init__(self, savefield=savefield, subsampling=subsampling)

    base.SecondOrderBackpropExtension = SecondOrderBackpropExtension
    hbp = _mod("backpack.extensions.secondorder.hbp")

    class LossHessianStrategy:
        EXACT = "exact"
        SAMPLING = "sampling"

    hbp.LossHessianStrategy = LossHessianStrategy
    sq = _mod("backpack.extensions.secondorder.sqrt_ggn")

    class _Any:
        def __init__(self, *a, **k):
            pass

    class _AnyModule(types.ModuleType):
        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)
            return _Any

    for sub in ["activations", "custom_module", "dropout", "flatten", "losses", "pad", "padding", "pooling", "slicing"]:
        m = _AnyModule(f"backpack.extensions.secondorder.sqrt_ggn.{sub}")
        sys.modules[m.__name__] = m
        setattr(sq, sub, m)
    mm = _mod("backpack.extensions.mat_to_mat_jac_base")

    class MatToJacMat:
        def __init__(self, derivatives, params=None):
            self.derivatives = derivatives

    mm.MatToJacMat = MatToJacMat
    _mod("backpack.core")
    _mod("backpack.core.derivatives")
    for sub, names in {
        "basederivatives": ["BaseDerivatives"],
        "linear": ["LinearDerivatives"],
        "conv1d": ["Conv1DDerivatives"],
        "conv2d": ["Conv2DDerivatives"],
        "conv3d": ["Conv3DDerivatives"],
        "conv_transpose1d": ["ConvTranspose1DDerivatives"],
        "conv_transpose2d": ["ConvTranspose2DDerivatives"],
        "conv_transpose3d": ["ConvTranspose3DDerivatives"],
        "batchnorm_nd": ["BatchNormNdDerivatives"],
    }.items():
        m = _mod(f"backpack.core.derivatives.{sub}")
        for n in names:
            setattr(m, n, type(n, (), {"__init__": lambda self, *a, **k: None}))
    _mod("backpack.custom_module")
    for sub, names in {
        "branching": ["SumModule", "Parallel"],
        "pad": ["Pad"],
        "scale_module": ["ScaleModule"],
        "slicing": ["Slicing"],
    }.items():
        m = _mod(f"backpack.custom_module.{sub}")
        for n in names:
            setattr(m, n, type(n, (torch.nn.Module,), {}))
    _mod("backpack.utils")
    ss = _mod("backpack.utils.subsampling")

    def subsample(tensor, dim=0, subsampling=None):
        if subsampling is None:
            return tensor
        return tensor.index_select(dim, torch.tensor(subsampling))

    ss.subsample = subsample
    hf = _mod("backpack.hessianfree")
    for sub in ["ggnvp", "hvp"]:
        m = _AnyModule(f"backpack.hessianfree.{sub}")
        sys.modules[m.__name__] = m
    cp = _AnyModule("backpack.utils.convert_parameters")
    sys.modules[cp.__name__] = cp
    return bp


def import_reference():
    _install_symeig_shim()
    _install_backpack_stub()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import vivit  # noqa: F401  (the real reference package)

    assert vivit.__file__.startswith(REF), vivit.__file__
    return vivit


class FakeModule(torch.nn.Module):
    """A leaf module holding parameters and the ``input0`` BackPACK stores (linalg/utils.py:54)."""

    def __init__(self, params, batch_size):
        super().__init__()
        for i, p in enumerate(params):
            self.register_parameter(f"p{i}", p)
        self.input0 = torch.zeros(batch_size, 1)


def make_factors(seed, C, N, shapes, N_grad=None, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    V = [torch.randn(C, N, *s, generator=g, dtype=dtype) / (N**0.5) for s in shapes]
    G = [torch.randn(N_grad or N, *s, generator=g, dtype=dtype) / (N_grad or N) for s in shapes]
    return V, G


def top_k_criterion(k, must_exceed=1e-5):
    """Same rule as the reference tests' make_criterion (test/optim/settings.py:21-47)."""

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


CASES = [
    # name, C, N, param shapes, N_grad, batch_size N_total, criterion, k
    dict(name="mlp_small", seed=1, C=3, N=4, shapes=[(5, 7), (5,)], k=2),
    dict(name="conv_like", seed=2, C=5, N=3, shapes=[(2, 3, 2, 2), (2,), (4, 6), (4,)], k=10),
    dict(name="subsampled", seed=3, C=4, N=2, shapes=[(6, 5), (6,)], k=3, N_total=6, N_grad=3),
    dict(name="mc1", seed=4, C=1, N=8, shapes=[(4, 9), (4,), (3, 4)], k=4),
    dict(name="wide", seed=5, C=10, N=6, shapes=[(12, 20), (12,)], k=10),
    # n = C*N = 320 > 192: the multi-kernel eigensolver (tridiagonalisation + D&C + back-transformation) instead of
    # the single-workgroup one that serves the reference's own test sizes
    dict(name="multikernel", seed=6, C=10, N=32, shapes=[(16, 24), (16,)], k=10),
]
# n = C*N = 2560 > 2048: the two-stage eigensolver the headline runs (band reduction, bulge chase, divide & conquer, Q2, Q1).
# The factors are 30 MB: only their SEED is stored (tests/helpers.py:planted_factors regenerates them in the test -- the build
# container and the GPU box run the same torch, whose CPU generator is deterministic) together with the reference's outputs.
BIG_CASES = [
    dict(name="two_stage", seed=8, C=10, N=256, shapes=[(48, 60), (48,)], k=10),
]


def run_case(vivit, case):
    from vivit.utils.ggn import Vmp
    from vivit.utils.gram import mVp, pairwise_dot, partial_contract

    C, N, shapes = case["C"], case["N"], case["shapes"]
    N_total = case.get("N_total", N)
    N_grad = case.get("N_grad", N_total)
    V, G = make_factors(case["seed"], C, N, shapes, N_grad=N_grad)
    out = {"C": C, "N": N, "N_total": N_total, "N_grad": N_grad, "k": case["k"]}
    for i, (v, g) in enumerate(zip(V, G)):
        out[f"V{i}"] = v.numpy()
        out[f"g{i}"] = g.numpy()
    subsampling = None if N_total == N else list(range(N))
    sub_grad = None if N_grad == N_total else list(range(N_grad))

    # --- raw contractions (K1, K2, K8, K9) straight from vivit/utils/{gram,ggn}.py
    out["gram_flat"] = sum(pairwise_dot(v, start_dim=2) for v in V).numpy()
    out["V_t_g0"] = partial_contract(V[0], G[0], (2, 1)).numpy()
    gen = torch.Generator().manual_seed(99)
    mat = torch.randn(3, C, N, generator=gen)
    out["mat"] = mat.numpy()
    out["Vmp0"] = Vmp(V[0], mat, 2).numpy()
    pm = torch.randn(2, *shapes[0], generator=gen)
    out["pmat"] = pm.numpy()
    out["mVp0"] = mVp(V[0], pm, 2).numpy()

    def fresh_params():
        return [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]

    # --- parameter-list forms: compute_gram_mat, V_mat_prod(concat), sqrt_gram_mat_prod(concat)
    from vivit.utils.ggn import V_mat_prod
    from vivit.utils.gram import compute_gram_mat, sqrt_gram_mat_prod

    params = fresh_params()
    for p_, v in zip(params, V):
        p_.sqrt_factor = v
    out["compute_gram_mat"] = compute_gram_mat(params, "sqrt_factor", 2, flatten=True).numpy()
    out["V_mat_prod_concat"] = V_mat_prod(mat, params, "sqrt_factor", concat=True).numpy()
    for i, r in enumerate(V_mat_prod(mat, params, "sqrt_factor")):
        out[f"V_mat_prod{i}"] = r.numpy()
    sub_idx = [N - 1, 0]  # sub-sampled application: V_t[:, subsampling] (vivit/utils/ggn.py:53-70)
    out["V_mat_prod_sub0"] = V_mat_prod(mat[:, :, :2].contiguous(), params, "sqrt_factor", subsampling=sub_idx)[0].numpy()
    cmat = torch.randn(C * N, 3, generator=gen)
    out["cmat"] = cmat.numpy()
    out["sqrt_gram_mat_prod_concat"] = sqrt_gram_mat_prod(cmat, params, "sqrt_factor", 2, concat=True).numpy()
    for i, r in enumerate(sqrt_gram_mat_prod(cmat, params, "sqrt_factor", 2)):
        out[f"sqrt_gram_mat_prod{i}"] = r.numpy()

    # --- EigvalshComputation / EighComputation (closures built from the reference's own utils)
    def attach_vivit(params, savefield):
        for p, v in zip(params, V):
            setattr(
                p,
                savefield,
                {
                    "gram_mat": (lambda v=v: pairwise_dot(v, start_dim=2, flatten=False)),
                    "V_mat_prod": (lambda m, v=v: Vmp(v, m, 2)),
                    "V_t_mat_prod": (lambda m, v=v: mVp(v, m, 2)),
                },
            )

    for groups_kind in ["one", "per_param"]:
        params = fresh_params()
        comp = vivit.EigvalshComputation(subsampling=subsampling)
        attach_vivit(params, "vivit_ggn_exact")
        groups = [{"params": params}] if groups_kind == "one" else [{"params": [p]} for p in params]
        hook = comp.get_extension_hook(groups)
        hook(FakeModule(params, N_total))
        for gi, grp in enumerate(groups):
            out[f"eigvalsh_{groups_kind}_{gi}"] = comp.get_result(grp).numpy()

    params = fresh_params()
    comp = vivit.EighComputation(subsampling=subsampling, warn_small_eigvals=0.0)
    attach_vivit(params, "vivit_ggn_exact")
    crit = top_k_criterion(case["k"])
    groups = [{"params": params, "criterion": crit}]
    comp.get_extension_hook(groups)(FakeModule(params, N_total))
    evals, evecs = comp.get_result(groups[0])
    out["eigh_evals"] = evals.numpy()
    for i, e in enumerate(evecs):
        out[f"eigh_evecs{i}"] = e.numpy()

    # --- DirectionalDerivatives / DirectionalDampedNewton on materialised factors
    mc = 1 if case["name"].startswith("mc") else 0
    savefield_ggn = "sqrt_ggn_mc" if mc else "sqrt_ggn_exact"

    def attach_sqrt(params):
        for p, v, g in zip(params, V, G):
            setattr(p, savefield_ggn, v.clone())
            p.grad_batch = g.clone()

    params = fresh_params()
    comp = vivit.DirectionalDerivativesComputation(
        subsampling_grad=sub_grad, subsampling_ggn=subsampling, mc_samples_ggn=mc, warn_small_eigvals=0.0
    )
    attach_sqrt(params)
    groups = [{"params": params, "criterion": crit}]
    comp.get_extension_hook(groups)(FakeModule(params, N_total))
    gammas, lambdas = comp.get_result(groups[0])
    out["gammas"] = gammas.numpy()
    out["lambdas"] = lambdas.numpy()

    params = fresh_params()
    comp = vivit.DirectionalDampedNewtonComputation(
        subsampling_grad=sub_grad, subsampling_ggn=subsampling, mc_samples_ggn=mc, warn_small_eigvals=0.0
    )
    attach_sqrt(params)
    groups = [{"params": params, "criterion": crit, "damping": constant_damping(1.0)}]
    comp.get_extension_hook(groups)(FakeModule(params, N_total))
    for i, s in enumerate(comp.get_result(groups[0])):
        out[f"newton{i}"] = s.numpy()

    # --- GramSqrtGGN hook (stand-alone Gram, K1 + accumulation)
    from vivit.extensions.hooks import GramBatchGrad, CenteredGramBatchGrad, GramSqrtGGNExact, GramSqrtGGNMC

    params = fresh_params()
    attach_sqrt(params)
    hook = (GramSqrtGGNMC if mc else GramSqrtGGNExact)()
    hook(FakeModule(params, N_total))
    out["gram_hook"] = hook.get_result().numpy()
    for cls, key in [(GramBatchGrad, "gram_batch_grad"), (CenteredGramBatchGrad, "gram_batch_grad_centered")]:
        params = fresh_params()
        attach_sqrt(params)
        hook = cls()
        hook(FakeModule(params, N_total))
        out[key] = hook.get_result().numpy()
    from vivit.extensions.hooks import CenteredBatchGrad

    params = fresh_params()
    attach_sqrt(params)
    hook = CenteredBatchGrad()
    hook(FakeModule(params, N_total))
    for i, p_ in enumerate(params):
        out[f"centered_grad_batch{i}"] = getattr(p_, hook.savefield).numpy()
    return out


def run_big_case(vivit, case):
    """EigvalshComputation, EighComputation (top k) and the directional derivatives / damped Newton step of the reference
    on seeded factors that are NOT stored (see BIG_CASES)."""
    from vivit.utils.ggn import Vmp
    from vivit.utils.gram import mVp, pairwise_dot

    sys.path.insert(0, os.path.dirname(OUT))
    from helpers import planted_factors   # the seeded recipe the tests regenerate the factors with (data, no product code)

    C, N, shapes = case["C"], case["N"], case["shapes"]
    V, G = planted_factors(case["seed"], C, N, shapes)
    out = {"C": C, "N": N, "N_total": N, "N_grad": N, "k": case["k"], "seed": case["seed"],
           "shapes_flat": np.array([d for s in shapes for d in (len(s),) + tuple(s)]),
           # fingerprints of the regenerated factors (a torch whose generator differs must fail loudly, not subtly)
           "V_checksum": np.array([float(v.double().sum()) for v in V] + [float(v.double().abs().sum()) for v in V]),
           "g_checksum": np.array([float(g.double().sum()) for g in G] + [float(g.double().abs().sum()) for g in G])}

    def fresh_params():
        return [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]

    def attach_vivit(params):
        for p, v in zip(params, V):
            p.vivit_ggn_exact = {"gram_mat": (lambda v=v: pairwise_dot(v, start_dim=2, flatten=False)),
                                 "V_mat_prod": (lambda m, v=v: Vmp(v, m, 2)), "V_t_mat_prod": (lambda m, v=v: mVp(v, m, 2))}

    def attach_sqrt(params):
        for p, v, g in zip(params, V, G):
            p.sqrt_ggn_exact = v.clone()
            p.grad_batch = g.clone()

    params = fresh_params()
    comp = vivit.EigvalshComputation()
    attach_vivit(params)
    groups = [{"params": params}]
    comp.get_extension_hook(groups)(FakeModule(params, N))
    out["eigvalsh_one_0"] = comp.get_result(groups[0]).numpy()

    crit = top_k_criterion(case["k"])
    params = fresh_params()
    comp = vivit.EighComputation(warn_small_eigvals=0.0)
    attach_vivit(params)
    groups = [{"params": params, "criterion": crit}]
    comp.get_extension_hook(groups)(FakeModule(params, N))
    evals, evecs = comp.get_result(groups[0])
    out["eigh_evals"] = evals.numpy()
    for i, e in enumerate(evecs):
        out[f"eigh_evecs{i}"] = e.numpy()

    params = fresh_params()
    comp = vivit.DirectionalDerivativesComputation(warn_small_eigvals=0.0)
    attach_sqrt(params)
    groups = [{"params": params, "criterion": crit}]
    comp.get_extension_hook(groups)(FakeModule(params, N))
    gammas, lambdas = comp.get_result(groups[0])
    out["gammas"], out["lambdas"] = gammas.numpy(), lambdas.numpy()

    params = fresh_params()
    comp = vivit.DirectionalDampedNewtonComputation(warn_small_eigvals=0.0)
    attach_sqrt(params)
    groups = [{"params": params, "criterion": crit, "damping": constant_damping(1.0)}]
    comp.get_extension_hook(groups)(FakeModule(params, N))
    for i, s in enumerate(comp.get_result(groups[0])):
        out[f"newton{i}"] = s.numpy()
    return out


def run_eig_utils(vivit):
    """vivit/utils/eig.py on the matrices of test/utils/test_stable_symeig.py:10-11 plus a PSD one."""
    from vivit.utils.eig import shift_diag, symeig, symeig_psd

    out = {}
    T1 = torch.diag(torch.Tensor([1.1, 2.2, 9.9]))
    T2 = torch.Tensor([[1.1, 2.2, 3.3], [4.4, 5.5, 6.6], [7.7, 8.8, 2.2]])
    g = torch.Generator().manual_seed(7)
    B = torch.randn(6, 3, generator=g)
    T3 = B @ B.T  # rank 3 of 6 -> zero eigenvalues to filter
    for name, T in [("T1", T1), ("T2", T2), ("T3", T3)]:
        out[name] = T.numpy()
        for shift in [0.0, 0.1, 1.0, 10.0]:
            w, v = symeig_psd(T.clone(), eigenvectors=True, shift=shift)
            out[f"{name}_psd_w_{shift}"] = w.numpy()
        w, v = symeig(T.clone(), eigenvectors=True)
        out[f"{name}_symeig_w"] = w.numpy()
        out[f"{name}_symeig_nvec"] = np.array(v.shape[1] if v.numel() else 0)
    inp = torch.tensor([[1.0, 1.0], [2.0, 2.0], [3.0, 4.0]])
    out["shift_nonsquare"] = shift_diag(inp, 0.1).numpy()
    return out


def main():
    vivit = import_reference()
    for case in CASES:
        res = run_case(vivit, case)
        np.savez_compressed(os.path.join(OUT, f"{case['name']}.npz"), **res)
        print(case["name"], {k: getattr(v, "shape", v) for k, v in list(res.items())[:4]}, "...", len(res), "arrays")
    for case in BIG_CASES:
        res = run_big_case(vivit, case)
        np.savez_compressed(os.path.join(OUT, f"{case['name']}.npz"), **res)
        print(case["name"], len(res), "arrays,", sum(v.nbytes for v in res.values() if hasattr(v, "nbytes")), "bytes")
    np.savez_compressed(os.path.join(OUT, "eig_utils.npz"), **run_eig_utils(vivit))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
