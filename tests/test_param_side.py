"""Parameter-side eigen-solve selection: a group with fewer parameters than Gram rows is decomposed
as the P x P GGN block V^T V instead of the n x n Gram matrix V V^T.  Results must agree with the reference
(Gram) path: identical non-zero spectrum, exact zeros where the Gram matrix is rank deficient, the same
parameter-space eigenvectors up to sign.  host flavour = Python layer on the oracle backend, hip = HIP kernels."""
import numpy as np
import pytest
import torch

import vivit_amd
from helpers import FakeModule, OracleBackend, set_kernel_backend, top_k_criterion
from vivit_amd import kernels
from vivit_amd.backend.extensions import _linear_weight_closures, _materialised_closures

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


def make_group(device, C, N, shapes, linear=False, seed=0):
    """Parameters with hand-made factors attached as closures; returns (params, attach)."""
    g = torch.Generator().manual_seed(seed)
    params = [torch.nn.Parameter(torch.zeros(*s, device=device)) for s in shapes]
    data = []
    for i, s in enumerate(shapes):
        if linear and len(s) == 2 and i == 0:
            sfac = torch.randn(C, N, s[0], generator=g).to(device) / N**0.5
            z = torch.randn(N, s[1], generator=g).to(device)
            data.append(("linear", sfac, z))
        else:
            data.append(("mat", torch.randn(C, N, *s, generator=g).to(device) / N**0.5))

    def attach(savefield):
        for p, d in zip(params, data):
            setattr(p, savefield, _linear_weight_closures(d[1], d[2]) if d[0] == "linear" else _materialised_closures(d[1]))

    return params, attach


@pytest.mark.parametrize("C,N,shapes,linear", [
    (3, 7, [(4, 5)], False),            # P = 20 < n = 21
    (2, 9, [(3, 2), (3,)], True),       # factorised Linear weight + bias, P = 9 < n = 18
    (3, 5, [(2, 2), (7,)], False),      # P = 11 < n = 15, two parameters
    (2, 4, [(6, 6)], False),            # P = 36 > n = 8: "auto" stays on the Gram side, "param" is forced
])
def test_eigvalsh_sides_agree(C, N, shapes, linear, device):
    res = {}
    for side in ["gram", "param", "auto"]:
        params, attach = make_group(device, C, N, shapes, linear)
        comp = vivit_amd.EigvalshComputation(side=side)
        attach(comp._savefield)
        group = {"params": params}
        comp.get_extension_hook([group])(FakeModule(params, N))
        res[side] = comp.get_result(group).detach().cpu().double().numpy()
        assert res[side].shape == (C * N,)
        assert all(not hasattr(p, comp._savefield) for p in params), "savefields must be deleted"
    scale = np.abs(res["gram"]).max()
    np.testing.assert_allclose(res["param"], res["gram"], rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(res["auto"], res["gram"], rtol=0, atol=2e-5 * scale)
    P = sum(int(np.prod(s)) for s in shapes)
    if P < C * N:  # exact zeros instead of rounding noise where the Gram matrix is rank deficient
        assert np.all(res["auto"][: C * N - P] == 0.0)


@pytest.mark.parametrize("C,N,shapes,linear,k", [
    (3, 7, [(4, 5)], False, 4),
    (2, 9, [(3, 2), (3,)], True, 3),
    (3, 5, [(2, 2), (7,)], False, 5),
    (2, 4, [(6, 6)], False, 3),
])
def test_eigh_sides_agree(C, N, shapes, linear, k, device):
    out = {}
    for side in ["gram", "param"]:
        params, attach = make_group(device, C, N, shapes, linear)
        comp = vivit_amd.EighComputation(side=side)
        attach(comp._savefield)
        group = {"params": params, "criterion": top_k_criterion(k)}
        comp.get_extension_hook([group])(FakeModule(params, N))
        evals, evecs = comp.get_result(group)
        assert [tuple(e.shape) for e in evecs] == [(k, *s) for s in shapes]
        flat = torch.cat([e.reshape(k, -1) for e in evecs], dim=1).detach().cpu().double().numpy()
        out[side] = (evals.detach().cpu().double().numpy(), flat)
        assert all(not hasattr(p, comp._savefield) for p in params)
    (wg, Vg), (wp, Vp) = out["gram"], out["param"]
    np.testing.assert_allclose(wp, wg, rtol=1e-4, atol=1e-5 * np.abs(wg).max())
    np.testing.assert_allclose(np.linalg.norm(Vp, axis=1), 1.0, atol=1e-4)   # unit norm over the group
    # same eigenvectors up to sign (the top-k eigenvalues of random factors are well separated)
    np.testing.assert_allclose(np.abs(np.sum(Vg * Vp, axis=1)), 1.0, atol=2e-3)


def test_eigh_param_side_zero_indices(device):
    """Indices that address the Gram matrix' exact zeros get a zero vector and the small-eigenvalue warning."""
    C, N, shapes = 3, 7, [(2, 3)]  # P = 6, n = 21
    params, attach = make_group(device, C, N, shapes)
    comp = vivit_amd.EighComputation(side="auto")
    attach(comp._savefield)
    group = {"params": params, "criterion": lambda evals: [0, len(evals) - 1]}
    with pytest.warns(UserWarning):
        comp.get_extension_hook([group])(FakeModule(params, N))
    evals, evecs = comp.get_result(group)
    assert float(evals[0]) == 0.0 and float(evals[1]) > 0
    assert float(evecs[0][0].abs().max()) == 0.0
    assert abs(float(evecs[0][1].norm()) - 1.0) < 1e-4


def test_side_argument_validated():
    with pytest.raises(ValueError):
        vivit_amd.EigvalshComputation(side="columns")
    with pytest.raises(ValueError):
        vivit_amd.EighComputation(side="")
