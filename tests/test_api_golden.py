"""The Computation classes / hooks of vivit_amd driven exactly like the reference's were when the
golden vectors were recorded (hand-made factors attached to parameters, ``hook(module)``).

Two flavours of every test:
  * ``hip``   (marked gpu): the product path, HIP kernels through the C ABI;
  * ``host``  (CPU): only the Python hook/scheduling/scaling layer, with the oracle standing in
    for the kernels (kernels.set_backend_for_testing) -- keeps the host logic covered without a GPU.
"""
import numpy as np
import pytest
import torch

import vivit_amd
from helpers import (
    CASES,
    FakeModule,
    OracleBackend,
    constant_damping,
    golden_factors,
    load_golden,
    top_k_criterion,
)
from vivit_amd import kernels
from vivit_amd.backend.extensions import _materialised_closures
from vivit_amd.extensions.hooks import CenteredGramBatchGrad, GramBatchGrad, GramSqrtGGNExact, GramSqrtGGNMC

FLAVOURS = [pytest.param("host", id="host"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=FLAVOURS)
def device(request):
    if request.param == "host":
        kernels.set_backend_for_testing(OracleBackend())
        yield torch.device("cpu")
        kernels.set_backend_for_testing(None)
    else:
        kernels.set_backend_for_testing(None)
        yield torch.device("cuda:0")


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b, rtol=rtol, atol=atol)


def fresh_params(V, device):
    return [torch.nn.Parameter(torch.zeros(*v.shape[2:], device=device)) for v in V]


def attach_vivit(params, V, savefield):
    for p, v in zip(params, V):
        setattr(p, savefield, _materialised_closures(v))


def attach_sqrt(params, V, G, savefield):
    for p, v, g in zip(params, V, G):
        setattr(p, savefield, v.clone())
        p.grad_batch = g.clone()


@pytest.mark.parametrize("case", CASES)
def test_eigvalsh(case, device):
    g = load_golden(case)
    V, _ = golden_factors(g, device)
    N, N_total = int(g["N"]), int(g["N_total"])
    sub = None if N == N_total else list(range(N))
    for kind in ["one", "per_param"]:
        params = fresh_params(V, device)
        comp = vivit_amd.EigvalshComputation(subsampling=sub)
        sf = comp._savefield
        attach_vivit(params, V, sf)
        groups = [{"params": params}] if kind == "one" else [{"params": [p]} for p in params]
        comp.get_extension_hook(groups)(FakeModule(params, N_total))
        for gi, grp in enumerate(groups):
            ref = g[f"eigvalsh_{kind}_{gi}"]
            # reference tolerance: rtol 1e-4, atol 5e-6 (test/linalg/test_eigvalsh.py:60) on the
            # top min(n, P) values; here on the whole spectrum with a norm-relative floor
            close(comp.get_result(grp), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
        assert all(not hasattr(p, sf) for p in params), "savefields must be deleted"


@pytest.mark.parametrize("case", CASES)
def test_eigh(case, device):
    g = load_golden(case)
    V, _ = golden_factors(g, device)
    N, N_total = int(g["N"]), int(g["N_total"])
    sub = None if N == N_total else list(range(N))
    params = fresh_params(V, device)
    comp = vivit_amd.EighComputation(subsampling=sub, warn_small_eigvals=0.0)
    attach_vivit(params, V, comp._savefield)
    groups = [{"params": params, "criterion": top_k_criterion(int(g["k"]))}]
    comp.get_extension_hook(groups)(FakeModule(params, N_total))
    evals, evecs = comp.get_result(groups[0])
    scale = np.abs(g["eigh_evals"]).max()
    close(evals, g["eigh_evals"], rtol=1e-4, atol=1e-5 * scale)  # test_eigh.py: 5e-4 / 1e-5
    for i, e in enumerate(evecs):
        assert e.shape == g[f"eigh_evecs{i}"].shape
        close(e.abs(), np.abs(g[f"eigh_evecs{i}"]), rtol=2e-2, atol=2e-3)  # sign-free, test_eigh.py:147-153
    # unit norm across the group (vivit/linalg/utils.py:67-76)
    sq = sum((e.flatten(1) ** 2).sum(1) for e in evecs)
    close(sq, np.ones(len(evals)), rtol=1e-5)


@pytest.mark.parametrize("case", CASES)
def test_directional_derivatives_and_newton(case, device):
    g = load_golden(case)
    V, G = golden_factors(g, device)
    N, N_total, N_grad = int(g["N"]), int(g["N_total"]), int(g["N_grad"])
    sub = None if N == N_total else list(range(N))
    sub_grad = None if N_grad == N_total else list(range(N_grad))
    mc = 1 if case.startswith("mc") else 0
    crit = top_k_criterion(int(g["k"]))

    comp = vivit_amd.DirectionalDerivativesComputation(
        subsampling_grad=sub_grad, subsampling_ggn=sub, mc_samples_ggn=mc, warn_small_eigvals=0.0
    )
    params = fresh_params(V, device)
    attach_sqrt(params, V, G, comp._savefield_ggn)
    groups = [{"params": params, "criterion": crit}]
    comp.get_extension_hook(groups)(FakeModule(params, N_total))
    gam, lam = comp.get_result(groups[0])
    # reference tolerances: gammas 1e-5/1e-4 (abs), lambdas 1e-5/1e-5
    close(gam.abs(), np.abs(g["gammas"]), rtol=1e-4, atol=1e-4 * np.abs(g["gammas"]).max())
    close(lam, g["lambdas"], rtol=1e-4, atol=1e-5 * np.abs(g["lambdas"]).max())
    assert all(not hasattr(p, comp._savefield_ggn) and not hasattr(p, "grad_batch") for p in params)

    comp = vivit_amd.DirectionalDampedNewtonComputation(
        subsampling_grad=sub_grad, subsampling_ggn=sub, mc_samples_ggn=mc, warn_small_eigvals=0.0
    )
    params = fresh_params(V, device)
    attach_sqrt(params, V, G, comp._savefield_ggn)
    groups = [{"params": params, "criterion": crit, "damping": constant_damping(1.0)}]
    comp.get_extension_hook(groups)(FakeModule(params, N_total))
    steps = comp.get_result(groups[0])
    for i, s in enumerate(steps):
        ref = g[f"newton{i}"]
        assert tuple(s.shape) == ref.shape
        close(s, ref, rtol=1e-4, atol=1e-5 * max(np.abs(ref).max(), 1e-3))  # test: 1e-5/1e-5


@pytest.mark.parametrize("case", CASES)
def test_gram_hooks(case, device):
    g = load_golden(case)
    V, G = golden_factors(g, device)
    mc = case.startswith("mc")
    sf = "sqrt_ggn_mc" if mc else "sqrt_ggn_exact"
    for layerwise in [False, True]:
        params = fresh_params(V, device)
        attach_sqrt(params, V, G, sf)
        hook = (GramSqrtGGNMC if mc else GramSqrtGGNExact)(layerwise=layerwise, free_sqrt_ggn=True)
        hook(FakeModule(params, int(g["N_total"])))
        close(hook.get_result(), g["gram_hook"], rtol=1e-5, atol=1e-6)
        assert all(not hasattr(p, sf) for p in params)
        assert all((getattr(p, hook.savefield) is not None) == layerwise for p in params)
    for cls, key in [(GramBatchGrad, "gram_batch_grad"), (CenteredGramBatchGrad, "gram_batch_grad_centered")]:
        params = fresh_params(V, device)
        attach_sqrt(params, V, G, sf)
        hook = cls()
        hook(FakeModule(params, int(g["N_total"])))
        close(hook.get_result(), g[key], rtol=1e-5, atol=1e-7)


def test_error_conventions(device):
    """ValueError / KeyError conventions of the reference."""
    from vivit_amd.utils.gram import partial_contract, sqrt_gram_mat_prod
    from vivit_amd.utils.hooks import ParameterGroupsHook

    p = torch.nn.Parameter(torch.zeros(2, device=device))
    for cls in [vivit_amd.EigvalshComputation, vivit_amd.EighComputation]:
        with pytest.raises(KeyError):
            cls().get_result({"params": []})
        with pytest.raises(ValueError):
            cls(subsampling=[0, 0, 1])
        with pytest.raises(ValueError):
            cls().get_extension_hook([{"no_params": []}])
    with pytest.raises(ValueError):
        vivit_amd.EighComputation().get_extension_hook([{"params": [p]}])  # no criterion
    with pytest.raises(ValueError):
        vivit_amd.DirectionalDampedNewtonComputation().get_extension_hook(
            [{"params": [p], "criterion": None}]
        )  # no damping
    with pytest.raises(ValueError):
        vivit_amd.EigvalshComputation().get_extension_hook([{"params": [p]}, {"params": [p]}])
    with pytest.raises(AssertionError):
        vivit_amd.DirectionalDampedNewtonComputation(mc_samples_ggn=2)
    with pytest.raises(ValueError):
        partial_contract(torch.zeros(2, 3, 4, device=device), torch.zeros(2, 3, device=device), (1, 1))
    with pytest.raises(NotImplementedError):
        sqrt_gram_mat_prod(torch.zeros(2, 3, 4, device=device), [], "x", 1)
    hook = ParameterGroupsHook([{"params": [p]}])
    with pytest.raises(ValueError):
        hook.get_output({"params": [p]})
