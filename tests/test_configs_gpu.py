"""BASELINE configs 3 and 5 at FULL size through the public API on one GPU, checked through size-independent
properties with plain autograd as the independent side (the reference's own method: GGN-vector products,
test/linalg/test_eigh.py:123-134; orthonormality :137-144; step == sum_k -gamma_k/(lambda_k+delta) e_k,
test/optim/test_directional_damped_newton.py:33-74; lambdas.mean(0) == evals,
docs/examples/basic_usage/example_directional_derivatives.py:192-199).  The oracle cannot run at these sizes."""
import math

import pytest
import torch
from torch import nn
from torch.func import functional_call, jvp

import vivit_amd
from helpers import constant_damping
from vivit_amd.backend import backpack, extend

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]


def top_k(k):
    def criterion(evals):
        n = evals.numel()
        return list(range(n - k, n))

    return criterion


def ggn_block_vec(model, lossname, X, params, vecs, samples=None):
    """``G_block @ v`` for the GGN block of ``params`` (mean loss) by autograd: J^T H (J v); ``vecs``: list (one entry
    per parameter) of ``[K, *p.shape]``.  ``samples`` [M, N, C]: the MC loss-Hessian sum_m s_m s_m^T / M instead."""
    names = {id(p): n for n, p in model.named_parameters()}
    sel = {names[id(p)]: p for p in params}
    rest = {n: p for n, p in model.named_parameters() if n not in sel}
    K = vecs[0].shape[0]
    out_cols = []
    for k in range(K):
        tang = {names[id(p)]: v[k] for p, v in zip(params, vecs)}

        def f(sp):
            return functional_call(model, {**rest, **sp}, (X,))

        out, Jv = jvp(f, (sel,), (tang,))
        N = out.shape[0]
        if samples is not None:
            p = out.softmax(1)
            S = (p.unsqueeze(0) - samples) / math.sqrt(samples.shape[0] * N)       # [M, N, C]
            t = (S * Jv.unsqueeze(0)).sum(2)                                        # [M, N]: s_mn . (J v)_n
            HJv = (S * t.unsqueeze(2)).sum(0)                                       # [N, C]: sum_m s_mn (s_mn . J_n v)
        elif lossname == "ce":
            p = out.softmax(1)
            HJv = (p * Jv - p * (p * Jv).sum(1, keepdim=True)) / N
        else:
            HJv = 2.0 * Jv / out.numel()
        out2 = functional_call(model, {**rest, **sel}, (X,))
        grads = torch.autograd.grad(out2, list(sel.values()), grad_outputs=HJv.detach())
        out_cols.append([g.detach() for g in grads])
    return [torch.stack([out_cols[k][i] for k in range(K)]) for i in range(len(params))]


def check_eigenpairs(model, lossname, X, params, evals, evecs, rtol=2e-3, samples=None):
    K = evals.numel()
    Gv = ggn_block_vec(model, lossname, X, params, evecs, samples=samples)
    E = torch.cat([e.reshape(K, -1) for e in evecs], 1).double()
    GE = torch.cat([g.reshape(K, -1) for g in Gv], 1).double()
    lam_max = evals.abs().max().item()
    res = (GE - evals.double()[:, None] * E).abs().max().item()
    assert res <= rtol * lam_max, f"eigen-residual {res / lam_max:.2e}"
    orth = (E @ E.T - torch.eye(K, device=E.device, dtype=E.dtype)).abs().max().item()
    assert orth <= 1e-3, f"orthonormality {orth:.2e}"


def lenet5():
    return nn.Sequential(
        nn.Conv2d(3, 6, 5), nn.ReLU(), nn.MaxPool2d(2), nn.Conv2d(6, 16, 5), nn.ReLU(), nn.MaxPool2d(2), nn.Flatten(),
        nn.Linear(400, 120), nn.ReLU(), nn.Linear(120, 84), nn.ReLU(), nn.Linear(84, 10))


def test_config3_lenet5_per_layer_blocks():
    """Config 3: LeNet-5 on CIFAR-10-shaped input, batch 2048, one group per layer (block-diagonal GGN), n = 20 480,
    P = 456 / 2 416 / 48 120 / 10 164 / 850; Gram side (the reference's path) and parameter side."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N = 2048
    model = lenet5().to(dev)
    X, y = torch.rand(N, 3, 32, 32, device=dev), torch.randint(0, 10, (N,), device=dev)
    layers = [m for m in model if len(list(m.parameters())) > 0]
    assert [sum(p.numel() for p in m.parameters()) for m in layers] == [456, 2416, 48120, 10164, 850]
    results = {}
    for side in ("gram", "auto"):
        model.zero_grad()
        m, lossf = extend(model), extend(nn.CrossEntropyLoss())
        groups = [{"params": list(layer.parameters()), "criterion": top_k(3)} for layer in layers]
        comp = vivit_amd.EighComputation(side=side)
        loss = lossf(m(X), y)
        with backpack(comp.get_extension(), extension_hook=comp.get_extension_hook(groups)):
            loss.backward()
        for gi, group in enumerate(groups):
            evals, evecs = comp.get_result(group)
            assert evals.shape == (3,) and bool((evals[1:] >= evals[:-1]).all())
            assert [tuple(e.shape) for e in evecs] == [(3, *p.shape) for p in group["params"]]
            check_eigenpairs(model, "ce", X, group["params"], evals, evecs)
            results[(side, gi)] = evals
    for gi in range(5):  # both sides see the same top spectrum
        a, b = results[("gram", gi)], results[("auto", gi)]
        assert (a - b).abs().max().item() <= 1e-4 * a.abs().max().item()
    # all eigenvalues of every block: rank <= min(n, P); both sides agree on the top min(n, P) (test_eigvalsh.py:55-60)
    spectra = {}
    for side in ("gram", "auto"):
        model.zero_grad()
        m, lossf = extend(model), extend(nn.CrossEntropyLoss())
        groups = [{"params": list(layer.parameters())} for layer in layers]
        comp = vivit_amd.EigvalshComputation(side=side)
        loss = lossf(m(X), y)
        with backpack(comp.get_extension(), extension_hook=comp.get_extension_hook(groups)):
            loss.backward()
        spectra[side] = [comp.get_result(g) for g in groups]
    for gi, layer in enumerate(layers):
        P = sum(p.numel() for p in layer.parameters())
        a, b = spectra["gram"][gi], spectra["auto"][gi]
        assert a.shape == b.shape == (10 * N,)
        k = min(10 * N, P)
        lam = a[-1].item()
        assert (a[-k:] - b[-k:]).abs().max().item() <= 1e-5 * lam + 5e-6
        if k < 10 * N:
            assert a[:-k].abs().max().item() <= 1e-5 * lam + 5e-6    # beyond the rank: rounding noise only
        assert (a[-3:] - results[("gram", gi)]).abs().max().item() <= 1e-5 * lam


def test_config5_wide_mlp_factorised_newton_step():
    """Config 5 on one GPU: MLP 4096-4096-1000, batch 32 768, MC (mc = 1) => n = 32 768, P = 20 878 312; the
    materialised factor would be 2.7 TB, so the factorised representation is the only one that exists.
    DirectionalDampedNewtonComputation(factorised=True), top-10 directions, damping 1."""
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info()
    if free < (120 << 30):
        pytest.skip("not enough free HBM")
    torch.manual_seed(0)
    N, K = 32768, 10
    model = nn.Sequential(nn.Linear(4096, 4096), nn.ReLU(), nn.Linear(4096, 1000)).to(dev)
    X, y = torch.rand(N, 4096, device=dev), torch.randint(0, 1000, (N,), device=dev)
    params = list(model.parameters())
    assert sum(p.numel() for p in params) == 20878312
    with torch.no_grad():
        probs = model(X).softmax(1)
        idx = torch.multinomial(probs, 1, replacement=True, generator=torch.Generator(device=dev).manual_seed(1))
        samples = torch.nn.functional.one_hot(idx.t(), 1000).float()          # [1, N, C]: shared by all three runs
        del probs

    def run(comp, exts, group):
        model.zero_grad()
        m, lossf = extend(model), extend(nn.CrossEntropyLoss())
        loss = lossf(m(X), y)
        with backpack(*exts, extension_hook=comp.get_extension_hook([group])):
            loss.backward()
        return comp.get_result(group)

    # (1) the step
    comp = vivit_amd.DirectionalDampedNewtonComputation(mc_samples_ggn=1, factorised=True)
    exts = comp.get_extensions()
    exts[1]._samples = samples
    group = {"params": params, "criterion": top_k(K), "damping": constant_damping(1.0)}
    step = run(comp, exts, group)
    assert [s.shape for s in step] == [p.shape for p in params]
    # (2) gammas / lambdas of the same directions
    comp = vivit_amd.DirectionalDerivativesComputation(mc_samples_ggn=1, factorised=True)
    exts = comp.get_extensions()
    exts[1]._samples = samples
    gam, lam = run(comp, exts, {"params": params, "criterion": top_k(K)})
    assert gam.shape == (N, K) and lam.shape == (N, K)
    # (3) the directions themselves in parameter space (ViViTGGNMC, factorised closures)
    comp = vivit_amd.EighComputation(mc_samples=1)
    ext = comp.get_extension()
    ext._samples = samples
    evals, evecs = run(comp, [ext], {"params": params, "criterion": top_k(K)})
    assert evals.shape == (K,) and bool((evals[1:] >= evals[:-1]).all()) and evals[0].item() > 0
    # lambdas.mean(0) == evals
    assert (lam.mean(0) - evals).abs().max().item() <= 1e-3 * evals[-1].item()
    # eigenpairs of the MC-GGN by autograd GGN-vector products
    check_eigenpairs(model, "ce", X, params, evals, evecs, samples=samples, rtol=5e-3)
    # step = sum_k -gamma_k / (lambda_k + 1) e_k: projections onto the directions (sign-free) and nothing outside them
    E = torch.cat([e.reshape(K, -1) for e in evecs], 1)
    s = torch.cat([t.reshape(-1) for t in step])
    coef = (E.double() @ s.double())
    expect = gam.double().mean(0).abs() / (lam.double().mean(0) + 1.0)
    assert (coef.abs() - expect).abs().max().item() <= 2e-3 * expect.max().item() + 1e-7
    assert abs(s.double().pow(2).sum().item() - coef.pow(2).sum().item()) <= 2e-3 * coef.pow(2).sum().item()
    # a descent direction of the mini-batch loss: g^T step = -sum_k gamma_k^2 / (lambda_k + 1) < 0
    model.zero_grad()
    nn.CrossEntropyLoss()(model(X), y).backward()
    g = torch.cat([p.grad.reshape(-1) for p in params])
    gs = (g.double() @ s.double()).item()
    target = -(gam.double().mean(0) ** 2 / (lam.double().mean(0) + 1.0)).sum().item()
    assert gs < 0 and abs(gs - target) <= 5e-3 * abs(target)


@pytest.mark.parametrize("N", [8, 1024])
def test_config4_resnet32_mc(N):
    """Config 4: ResNet-32, CIFAR-100-shaped (C = 100), BatchNorm in eval mode, residual adds, SqrtGGN-MC with one
    sample => n = N, P = 470 004.  N = 8: the Gram matrix against brute-force autograd (per-sample Jacobian rows of the
    MC factor); N = 1024 (BASELINE): eigenpairs against autograd MC-GGN-vector products, all eigenvalues two ways."""
    from helpers import resnet32

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = resnet32(100).to(dev)
    params = list(model.parameters())
    assert sum(p.numel() for p in params) == 470004
    X, y = torch.rand(N, 3, 32, 32, device=dev), torch.randint(0, 100, (N,), device=dev)
    with torch.no_grad():
        probs = model(X).softmax(1)
        idx = torch.multinomial(probs, 1, replacement=True, generator=torch.Generator(device=dev).manual_seed(2))
        samples = torch.nn.functional.one_hot(idx.t(), 100).float()           # [1, N, C]

    def run(comp, group):
        model.zero_grad()
        m, lossf = extend(model), extend(nn.CrossEntropyLoss())
        ext = comp.get_extension()
        ext._samples = samples
        loss = lossf(m(X), y)
        with backpack(ext, extension_hook=comp.get_extension_hook([group])):
            loss.backward()
        return comp.get_result(group)

    K = min(10, N)
    evals, evecs = run(vivit_amd.EighComputation(mc_samples=1), {"params": params, "criterion": top_k(K)})
    assert evals.shape == (K,) and evals[0].item() > 0
    check_eigenpairs(model, "ce", X, params, evals, evecs, samples=samples, rtol=5e-3)
    all_evals = run(vivit_amd.EigvalshComputation(mc_samples=1), {"params": params})
    assert all_evals.shape == (N,)
    assert (all_evals[-K:] - evals).abs().max().item() <= 1e-4 * evals[-1].item()
    if N <= 8:
        # brute force: V_t[0, n, :] = S[0, n, :] J_n, row by row with autograd; Gram spectrum must match
        out = model(X)
        S = ((out.softmax(1).detach().unsqueeze(0) - samples) / math.sqrt(N))[0]   # [N, C]
        rows = []
        for n_ in range(N):
            grads = torch.autograd.grad((out[n_] * S[n_]).sum(), params, retain_graph=True)
            rows.append(torch.cat([g.reshape(-1) for g in grads]))
        A = torch.stack(rows).double()
        ref = torch.linalg.eigvalsh(A @ A.T)
        assert (all_evals.double() - ref).abs().max().item() <= 1e-4 * ref[-1].item()
