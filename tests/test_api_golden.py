"""The Computation classes / hooks of vivit_amd driven exactly like the reference's were when the
golden vectors were recorded (hand-made factors attached to parameters, ``hook(module)``).

Two flavours of every test:
  * ``hip``   (marked gpu): the product path, HIP kernels through the C ABI;
  * ``host``  (CPU): only the Python hook/scheduling/scaling layer, with the oracle standing in
    for the kernels (helpers.set_kernel_backend monkeypatches the launchers) -- keeps the host logic covered without a GPU.
"""
import numpy as np
import pytest
import torch

import vivit_amd
from helpers import (
    BIG_CASES,
    CASES,
    FakeModule,
    OracleBackend,
    set_kernel_backend,
    constant_damping,
    golden_factors,
    load_golden,
    top_k_criterion,
)
from vivit_amd import kernels
from vivit_amd.backend.extensions import _materialised_closures
from vivit_amd.extensions.hooks import (
    CenteredBatchGrad,
    CenteredGramBatchGrad,
    GramBatchGrad,
    GramSqrtGGNExact,
    GramSqrtGGNMC,
)

FLAVOURS = [pytest.param("host", id="host"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=FLAVOURS)
def device(request):
    if request.param == "host":
        set_kernel_backend(OracleBackend())
        yield torch.device("cpu")
        set_kernel_backend(None)
    else:
        set_kernel_backend(None)
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


@pytest.mark.parametrize("case", BIG_CASES)
def test_two_stage_size_against_the_reference(case, device):
    """the solver the headline runs (n > 2048: band reduction, bulge chase, divide & conquer, Q2, Q1) pinned
    by outputs of the IMPORTED REFERENCE, not only by fp64 LAPACK: n = C N = 2560, P = 2928 (so the sample side is the smaller
    one), ``EigvalshComputation`` (whole spectrum), ``EighComputation`` top-10, gammas / lambdas and the damped Newton step, at
    the reference tests' tolerances (test/linalg/test_eigvalsh.py:55-60, test_eigh.py:147-153,
    test/optim/test_directional_damped_newton.py:72).  The 30 MB of factors are regenerated from their seed."""
    g = load_golden(case)
    V, G = golden_factors(g, device)
    N = int(g["N"])
    crit = top_k_criterion(int(g["k"]))
    assert V[0].shape[0] * N == 2560

    params = fresh_params(V, device)
    comp = vivit_amd.EigvalshComputation()
    attach_vivit(params, V, comp._savefield)
    groups = [{"params": params}]
    comp.get_extension_hook(groups)(FakeModule(params, N))
    ref = g["eigvalsh_one_0"]
    scale = np.abs(ref).max()
    close(comp.get_result(groups[0]), ref, rtol=1e-4, atol=1e-5 * scale)

    params = fresh_params(V, device)
    comp = vivit_amd.EighComputation(warn_small_eigvals=0.0)
    attach_vivit(params, V, comp._savefield)
    groups = [{"params": params, "criterion": crit}]
    comp.get_extension_hook(groups)(FakeModule(params, N))
    evals, evecs = comp.get_result(groups[0])
    close(evals, g["eigh_evals"], rtol=1e-4, atol=1e-5 * scale)
    for i, e in enumerate(evecs):
        assert e.shape == g[f"eigh_evecs{i}"].shape
        close(e.abs(), np.abs(g[f"eigh_evecs{i}"]), rtol=2e-2, atol=2e-3)

    comp = vivit_amd.DirectionalDerivativesComputation(warn_small_eigvals=0.0)
    params = fresh_params(V, device)
    attach_sqrt(params, V, G, comp._savefield_ggn)
    groups = [{"params": params, "criterion": crit}]
    comp.get_extension_hook(groups)(FakeModule(params, N))
    gam, lam = comp.get_result(groups[0])
    close(gam.abs(), np.abs(g["gammas"]), rtol=1e-4, atol=1e-4 * np.abs(g["gammas"]).max())
    close(lam, g["lambdas"], rtol=1e-4, atol=1e-5 * np.abs(g["lambdas"]).max())

    comp = vivit_amd.DirectionalDampedNewtonComputation(warn_small_eigvals=0.0)
    params = fresh_params(V, device)
    attach_sqrt(params, V, G, comp._savefield_ggn)
    groups = [{"params": params, "criterion": crit, "damping": constant_damping(1.0)}]
    comp.get_extension_hook(groups)(FakeModule(params, N))
    for i, s in enumerate(comp.get_result(groups[0])):
        ref = g[f"newton{i}"]
        assert tuple(s.shape) == ref.shape
        close(s, ref, rtol=1e-4, atol=1e-5 * max(np.abs(ref).max(), 1e-3))


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
        close(hook.get_result(), g["gram_hook"], rtol=1e-5, atol=5e-6)   # (entries up to ~45: summation-order noise of the host BLAS)
        assert all(not hasattr(p, sf) for p in params)
        assert all((getattr(p, hook.savefield) is not None) == layerwise for p in params)
    for cls, key in [(GramBatchGrad, "gram_batch_grad"), (CenteredGramBatchGrad, "gram_batch_grad_centered")]:
        params = fresh_params(V, device)
        attach_sqrt(params, V, G, sf)
        hook = cls()
        hook(FakeModule(params, int(g["N_total"])))
        close(hook.get_result(), g[key], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("case", CASES)
def test_contractions_and_parameter_list_forms(case, device):
    """vivit_amd.utils.{gram,ggn} against what the reference's own functions returned (K1, K2, K8, K9):
    partial_contract / Vmp / mVp on one factor, compute_gram_mat / V_mat_prod / sqrt_gram_mat_prod over the
    parameter list (vivit/utils/gram.py:72-179,182-203, vivit/utils/ggn.py:11-115)."""
    from vivit_amd.utils.ggn import V_mat_prod, V_param_mat_prod, Vmp
    from vivit_amd.utils.gram import compute_gram_mat, mVp, pairwise_dot, partial_contract, sqrt_gram_mat_prod

    g = load_golden(case)
    V, G = golden_factors(g, device)
    scale = np.abs(g["gram_flat"]).max()
    mat = torch.from_numpy(g["mat"]).to(device)
    pmat = torch.from_numpy(g["pmat"]).to(device)
    cmat = torch.from_numpy(g["cmat"]).to(device)
    close(partial_contract(V[0], G[0], (2, 1)), g["V_t_g0"], rtol=1e-5, atol=1e-6 * scale)
    close(Vmp(V[0], mat, 2), g["Vmp0"], rtol=1e-5, atol=1e-5)
    close(mVp(V[0], pmat, 2), g["mVp0"], rtol=1e-5, atol=1e-5)
    close(sum(pairwise_dot(v, start_dim=2) for v in V), g["gram_flat"], rtol=1e-5, atol=1e-6 * scale)

    params = fresh_params(V, device)
    for p, v in zip(params, V):
        p.sqrt_factor = v
    gram = compute_gram_mat(params, "sqrt_factor", 2, flatten=True)
    assert gram.shape == g["compute_gram_mat"].shape
    close(gram, g["compute_gram_mat"], rtol=1e-5, atol=1e-6 * scale)
    C, N = V[0].shape[:2]
    assert compute_gram_mat(params, "sqrt_factor", 2, flatten=False).shape == (C, N, C, N)
    for i, r in enumerate(V_mat_prod(mat, params, "sqrt_factor")):
        assert tuple(r.shape) == g[f"V_mat_prod{i}"].shape
        close(r, g[f"V_mat_prod{i}"], rtol=1e-5, atol=1e-5)
    close(V_mat_prod(mat, params, "sqrt_factor", concat=True), g["V_mat_prod_concat"], rtol=1e-5, atol=1e-5)
    sub = V_param_mat_prod(params[0], mat[:, :, :2].contiguous(), "sqrt_factor", subsampling=[N - 1, 0])
    close(sub, g["V_mat_prod_sub0"], rtol=1e-5, atol=1e-5)
    with pytest.raises(AssertionError):
        V_mat_prod(mat[0], params, "sqrt_factor")
    for i, r in enumerate(sqrt_gram_mat_prod(cmat, params, "sqrt_factor", 2)):
        assert tuple(r.shape) == g[f"sqrt_gram_mat_prod{i}"].shape
        close(r, g[f"sqrt_gram_mat_prod{i}"], rtol=1e-5, atol=1e-5)
    close(sqrt_gram_mat_prod(cmat, params, "sqrt_factor", 2, concat=True), g["sqrt_gram_mat_prod_concat"],
          rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", CASES)
def test_centered_batch_grad(case, device):
    """CenteredBatchGrad (vivit/extensions/firstorder/batch_grad/gram_batch_grad.py:7-37)."""
    g = load_golden(case)
    V, G = golden_factors(g, device)
    params = fresh_params(V, device)
    attach_sqrt(params, V, G, "sqrt_ggn_exact")
    hook = CenteredBatchGrad()
    hook(FakeModule(params, int(g["N_total"])))
    for i, p in enumerate(params):
        close(getattr(p, hook.savefield), g[f"centered_grad_batch{i}"], rtol=1e-5, atol=1e-7)


def test_eig_utils(device):
    """vivit_amd.utils.eig against the reference's vivit/utils/eig.py on the matrices of
    test/utils/test_stable_symeig.py:10-11 (T2 is NOT symmetric: ``upper=True`` must read its upper triangle)."""
    from vivit_amd.utils.eig import remove_zero_evals, shift_diag, symeig, symeig_psd

    g = load_golden("eig_utils")
    for name in ["T1", "T2", "T3"]:
        T = torch.from_numpy(g[name]).to(device)
        for shift in [0.0, 0.1, 1.0, 10.0]:
            for inplace in [False, True]:
                inp = T.clone()
                w, v = symeig_psd(inp, eigenvectors=True, shift=shift, shift_inplace=inplace)
                close(w, g[f"{name}_psd_w_{shift}"], rtol=1e-5, atol=1e-5)
                # the input is never destroyed; an in-place shift is taken back (test_stable_symeig.py:48-74)
                close(inp, g[name], rtol=1e-6, atol=1e-6)
                assert v.shape == T.shape
            w_only, empty = symeig_psd(T.clone(), eigenvectors=False, shift=shift)
            close(w_only, g[f"{name}_psd_w_{shift}"], rtol=1e-5, atol=1e-5)
            assert empty.numel() == 0
        w, v = symeig(T.clone(), eigenvectors=True)
        # fp32 rounding noise of the null space is ~1e-7 * lambda_max: absolute floor scaled like test_eigvalsh.py:55-60
        close(w, g[f"{name}_symeig_w"], rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(g[f"{name}_symeig_w"]).max())))
        assert (v.shape[1] if v.numel() else 0) == int(g[f"{name}_symeig_nvec"])
    # eigenvectors of the symmetric cases: T v = w v with the upper triangle as the matrix
    T3 = torch.from_numpy(g["T3"]).to(device)
    w, v = symeig_psd(T3.clone(), eigenvectors=True)
    close(T3 @ v, (v * w).cpu().numpy(), rtol=1e-4, atol=1e-5)
    w, v = remove_zero_evals(w, v, atol=1e-4)  # rank 3 of 6
    assert w.numel() == 3 and v.shape == (6, 3)
    inp = torch.tensor([[1.0, 1.0], [2.0, 2.0], [3.0, 4.0]], device=device)
    close(shift_diag(inp, 0.1), g["shift_nonsquare"])  # non-square input (test_stable_symeig.py:104-121)
    assert shift_diag(inp, 0.0) is inp
    with pytest.raises(ValueError):
        symeig_psd(torch.zeros(2, 2, 2, device=device))
    with pytest.raises(ValueError):
        symeig(torch.zeros(4, device=device))


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
