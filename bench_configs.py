"""The other BASELINE configurations (1, 3, 4, 5 on one GPU) through the PUBLIC API, on real factors from the stand-in
BackPACK backend (random-init models of the named architectures, uniform random inputs) -- `bench.py` attaches the
result as its ``configs`` block ("plus configs 1, 3, 4, 5").

Every line is timed three ways so that the share of the factor provider is visible:
  backward_s         plain ``loss.backward()`` (autograd only)
  factors_s          ``with backpack(<extension>): loss.backward()`` -- the backward pass incl. the sqrt-GGN factor
                     back-propagation of the extension the Computation asks for, no hook
  total_s            the same with the Computation's extension hook: factors + Gram + eigensolver (+ criterion callback,
                     back-projection and normalisation for ``eigh``; + gammas / lambdas / step for the Newton step)
  path_s             total_s - factors_s: the hot path itself (what the reference spends in vivit/utils/gram.py,
                     Tensor.symeig and the einsums behind them)
Median of ``reps`` (3) after one warm-up, wall clock around ``torch.cuda.synchronize()``.
"""
import time

import torch
from torch import nn


def lenet5():
    """LeNet-5 on CIFAR-10-shaped input: 456 / 2 416 / 48 120 / 10 164 / 850 parameters per layer (BASELINE config 3)."""
    return nn.Sequential(
        nn.Conv2d(3, 6, 5), nn.ReLU(), nn.MaxPool2d(2), nn.Conv2d(6, 16, 5), nn.ReLU(), nn.MaxPool2d(2), nn.Flatten(),
        nn.Linear(400, 120), nn.ReLU(), nn.Linear(120, 84), nn.ReLU(), nn.Linear(84, 10))


def resnet32(num_classes=100):
    """CIFAR ResNet-32 (3 stages x 5 basic blocks, 16/32/64 channels, option-A shortcuts: stride-2 sub-sampling and
    zero-padded channels), written with the branching modules so that every operation is a leaf module.
    470 004 parameters for 100 classes (BASELINE config 4)."""
    from vivit_amd.backend import ActiveIdentity, Pad, Parallel, Slicing

    def block(cin, cout, stride):
        body = nn.Sequential(
            nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(),
            nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout))
        if stride == 1 and cin == cout:
            shortcut = ActiveIdentity()
        else:
            pad = (cout - cin) // 2
            shortcut = nn.Sequential(Slicing((slice(None), slice(None), slice(None, None, 2), slice(None, None, 2))),
                                     Pad((0, 0, 0, 0, pad, pad)))
        return nn.Sequential(Parallel(shortcut, body), nn.ReLU())

    layers = [nn.Conv2d(3, 16, 3, padding=1, bias=False), nn.BatchNorm2d(16), nn.ReLU()]
    cin = 16
    for cout, stride in ((16, 1), (32, 2), (64, 2)):
        for b in range(5):
            layers.append(block(cin, cout, stride if b == 0 else 1))
            cin = cout
    layers += [nn.AvgPool2d(8), nn.Flatten(), nn.Linear(64, num_classes)]
    model = nn.Sequential(*layers)
    g = torch.Generator().manual_seed(0)
    for m in model.modules():  # non-trivial running statistics (eval mode, as in the reference tests)
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.copy_(torch.rand(m.num_features, generator=g) - 0.5)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    return model.eval()


def top_k(k):
    def criterion(evals):
        n = evals.numel()
        return list(range(max(n - k, 0), n))

    return criterion


def _median(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[(len(ts) - 1) // 2]   # median (the lower one of an even count: a single hiccup must not set the line)


class _Problem:
    """A model + batch on the device, and the three timed variants of its backward pass."""

    def __init__(self, model, X, y, samples=None):
        from vivit_amd.backend import extend

        self.model, self.X, self.y, self.samples = extend(model), X, y, samples
        self.lossf = extend(nn.CrossEntropyLoss())

    def backward(self, extensions=(), hook=None):
        from vivit_amd.backend import backpack

        self.model.zero_grad(set_to_none=True)
        loss = self.lossf(self.model(self.X), self.y)
        for e in extensions:
            if self.samples is not None and hasattr(e, "_samples"):
                e._samples = self.samples
        if extensions:
            with backpack(*extensions, extension_hook=hook):
                loss.backward()
        else:
            loss.backward()

    def line(self, name, make, pairs, reps, note):
        """``make()`` -> (computation, extensions, groups); ``pairs``: eigenpairs (or eigenvalues) one pass delivers."""
        t_bwd = _median(lambda: self.backward(), reps)

        def factors_only():
            comp, exts, groups = make()
            self.backward(exts)
            for p in self.model.parameters():          # drop the saved factors (the hook would have consumed them)
                for e in exts:
                    if hasattr(p, e.savefield):
                        delattr(p, e.savefield)

        def total():
            comp, exts, groups = make()
            self.backward(exts, comp.get_extension_hook(groups))
            return comp, groups

        t_fac = _median(factors_only, reps)
        t_tot = _median(total, reps)
        path = max(t_tot - t_fac, 1e-9)
        return {"config": name, "backward_s": t_bwd, "factors_s": t_fac, "total_s": t_tot, "path_s": path,
                "pairs": pairs, "pairs_per_s_total": pairs / t_tot, "pairs_per_s_path": pairs / path,
                "factor_share": t_fac / t_tot, "note": note}


def run_configs(device, reps=3, which=("1", "3", "4", "5"), progress=lambda m: None):
    """Returns the list of config lines; each config is skipped (with the reason) rather than failing the benchmark."""
    import vivit_amd

    out = []

    def guarded(tag, fn):
        try:
            torch.cuda.empty_cache()
            fn()
        except Exception as exc:  # noqa: BLE001 -- a secondary line must never take the headline down
            out.append({"config": tag, "error": repr(exc)})
        progress(f"configs: {tag} done")

    def cfg1():
        torch.manual_seed(0)
        N = 128
        model = nn.Sequential(nn.Linear(784, 512), nn.ReLU(), nn.Linear(512, 10)).to(device)
        pb = _Problem(model, torch.rand(N, 784, device=device), torch.randint(0, 10, (N,), device=device))
        params = list(model.parameters())
        n = 10 * N

        def eigvalsh():
            comp = vivit_amd.EigvalshComputation()
            return comp, [comp.get_extension()], [{"params": params}]

        def eigh_all():
            comp = vivit_amd.EighComputation(warn_small_eigvals=0.0)
            return comp, [comp.get_extension()], [{"params": params, "criterion": top_k(n)}]

        def eigh10():
            comp = vivit_amd.EighComputation()
            return comp, [comp.get_extension()], [{"params": params, "criterion": top_k(10)}]

        note = "MLP 784-512-10, N = 128, exact GGN, one group: n = 1280, P = 407 050 (Linear weights factorised, linear.py:41-81)"
        out.append(pb.line("1: eigvalsh", eigvalsh, n, reps, note))
        out.append(pb.line("1: eigh, all 1280 eigenvectors in parameter space", eigh_all, n, reps, note))
        out.append(pb.line("1: eigh, criterion top-10", eigh10, 10, reps, note))

    def cfg3():
        torch.manual_seed(0)
        N = 2048
        model = lenet5().to(device)
        pb = _Problem(model, torch.rand(N, 3, 32, 32, device=device), torch.randint(0, 10, (N,), device=device))
        layers = [m for m in model if len(list(m.parameters())) > 0]
        n = 10 * N
        note = "LeNet-5 CIFAR-10-shaped, N = 2048, exact GGN, one group per layer (block-diagonal GGN): n = 20 480, P = 456/2416/48120/10164/850"
        for side in ("gram", "auto"):
            def eigvalsh(side=side):
                comp = vivit_amd.EigvalshComputation(side=side)
                return comp, [comp.get_extension()], [{"params": list(layer.parameters())} for layer in layers]

            def eigh10(side=side):
                comp = vivit_amd.EighComputation(side=side)
                return comp, [comp.get_extension()], [{"params": list(layer.parameters()), "criterion": top_k(10)} for layer in layers]

            tag = "Gram side (the reference's path)" if side == "gram" else "side='auto' (parameter side where P < n)"
            r3 = 1 if side == "gram" else reps   # (2-3 s per pass on the Gram side: one timed pass behind the warm-up)
            out.append(pb.line(f"3: eigvalsh, 5 blocks, {tag}", eigvalsh, 5 * n, r3, note))
            out.append(pb.line(f"3: eigh top-10 per block, {tag}", eigh10, 50, r3, note))

    def cfg4():
        torch.manual_seed(0)
        N = 1024
        model = resnet32(100).to(device)
        X, y = torch.rand(N, 3, 32, 32, device=device), torch.randint(0, 100, (N,), device=device)
        with torch.no_grad():
            idx = torch.multinomial(model(X).softmax(1), 1, replacement=True, generator=torch.Generator(device=device).manual_seed(2))
            samples = torch.nn.functional.one_hot(idx.t(), 100).float()
        pb = _Problem(model, X, y, samples)
        params = list(model.parameters())
        note = "ResNet-32 CIFAR-100-shaped (C = 100), SqrtGGN-MC mc = 1, N = 1024, one group: n = 1024, P = 470 004"

        def eigvalsh():
            comp = vivit_amd.EigvalshComputation(mc_samples=1)
            return comp, [comp.get_extension()], [{"params": params}]

        def eigh10():
            comp = vivit_amd.EighComputation(mc_samples=1)
            return comp, [comp.get_extension()], [{"params": params, "criterion": top_k(10)}]

        out.append(pb.line("4: eigvalsh", eigvalsh, N, reps, note))
        out.append(pb.line("4: eigh, criterion top-10", eigh10, 10, reps, note))

    def cfg5():
        free, _ = torch.cuda.mem_get_info()
        if free < (120 << 30):
            out.append({"config": "5", "skipped": "needs 120 GB of free HBM"})
            return
        torch.manual_seed(0)
        N = 32768
        model = nn.Sequential(nn.Linear(4096, 4096), nn.ReLU(), nn.Linear(4096, 1000)).to(device)
        X, y = torch.rand(N, 4096, device=device), torch.randint(0, 1000, (N,), device=device)
        with torch.no_grad():
            idx = torch.multinomial(model(X).softmax(1), 1, replacement=True, generator=torch.Generator(device=device).manual_seed(1))
            samples = torch.nn.functional.one_hot(idx.t(), 1000).float()
        pb = _Problem(model, X, y, samples)
        params = list(model.parameters())
        note = ("wide MLP 4096-4096-1000, N = 32 768 on ONE GPU, MC mc = 1, factorised Linear weights: n = 32 768, P = 20 878 312; "
                "DirectionalDampedNewtonComputation(factorised=True), top-10 directions, damping 1")

        def newton():
            comp = vivit_amd.DirectionalDampedNewtonComputation(mc_samples_ggn=1, factorised=True)
            group = {"params": params, "criterion": top_k(10), "damping": lambda ev, evecs, g, l: torch.ones_like(ev)}
            return comp, comp.get_extensions(), [group]

        out.append(pb.line("5 (single GPU): damped Newton step, top-10", newton, 10, 1, note))

    fns = {"1": cfg1, "3": cfg3, "4": cfg4, "5": cfg5}
    for k in which:
        guarded(k, fns[k])
    return out


if __name__ == "__main__":
    import json
    import sys

    dev = torch.device("cuda:0")
    res = run_configs(dev, which=tuple(sys.argv[1:]) or ("1", "3", "4", "5"), progress=lambda m: print(m, file=sys.stderr, flush=True))
    for r in res:
        print(json.dumps(r))
