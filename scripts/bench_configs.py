"""Group eigenproblems of the other BASELINE configurations with synthetic factors of
the right shapes: Gram build + symeig with eigenvectors per parameter group, Gram side vs parameter side.
  cfg1  MLP 784-512-10, N=128:   n = 1280,  P = 407050 (one group)
  cfg3  LeNet-5 CIFAR-10, N=2048: n = 20480, five groups P = 456 / 2416 / 48120 / 10164 / 850
  cfg4  ResNet-32 (MC, M=1), N=1024: n = 1024, P = 470004 (one group)
  cfg5  MLP 4096-4096-1000 (MC), N=32768: n = 32768, factorised Linear Gram (two layers), one group
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vivit_amd import kernels

dev = torch.device("cuda:0")


def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def gram(V):
    """Gram matrix of one factor, or of a list of per-parameter factors accumulated in-kernel (beta = 1) like the hooks do."""
    if not isinstance(V, (list, tuple)):
        return kernels.gram_syrk(V)
    G = kernels.gram_syrk(V[0])
    for Vp in V[1:]:
        kernels.gram_syrk(Vp, out=G, beta=1.0)
    return G


def gram_side(V):
    G = gram(V)
    return kernels.symeig(G, eigenvectors=True, overwrite=True)


def gram_side_top10(V):
    G = gram(V)
    plan = kernels.symeig_reduce(G, overwrite=True)
    n = plan.n
    return plan.evals, plan.select(list(range(n - 10, n)))


def param_side(V):
    H = kernels.gemm_tn(V, V)
    return kernels.symeig(H, eigenvectors=True, overwrite=True)


which = sys.argv[1:] or ["cfg1", "cfg3", "cfg4", "cfg5"]
if "cfg1" in which:
    # the four parameters of the MLP (weight 512 x 784, bias 512, weight 10 x 512, bias 10), one factor each as in the API
    V = [torch.randn(1280, p, device=dev) / 128**0.5 for p in (401408, 512, 5120, 10)]
    t, _ = timed(lambda: gram_side(V))
    t10, _ = timed(lambda: gram_side_top10(V))
    print(f"cfg1 n=1280 P=407050: Gram+symeig(vectors) {t*1e3:.1f} ms -> {1280/t:.0f} eigenpairs/s; top-10: {t10*1e3:.1f} ms")
    del V
if "cfg4" in which:
    V = torch.randn(1024, 470004, device=dev) / 1024**0.5
    t, _ = timed(lambda: gram_side(V))
    t10, _ = timed(lambda: gram_side_top10(V))
    print(f"cfg4 n=1024 P=470004: Gram+symeig(vectors) {t*1e3:.1f} ms -> {1024/t:.0f} eigenpairs/s; top-10: {t10*1e3:.1f} ms")
    del V
if "cfg3" in which:
    n, tot_g, tot_p, tot_10 = 20480, 0.0, 0.0, 0.0
    for P in [456, 2416, 48120, 10164, 850]:
        V = torch.randn(n, P, device=dev) / 2048**0.5
        tg, _ = timed(lambda: gram_side(V), reps=1)
        t10, _ = timed(lambda: gram_side_top10(V), reps=1)
        tot_10 += t10
        if P < n:
            tp, _ = timed(lambda: param_side(V), reps=1)
        else:
            tp = tg
        tot_g += tg; tot_p += tp
        print(f"cfg3 group P={P}: Gram side {tg*1e3:.1f} ms, parameter side {tp*1e3:.1f} ms")
        del V
    print(f"cfg3 total (5 groups, n=20480): Gram side {tot_g:.2f} s, auto side {tot_p:.2f} s, Gram side top-10 {tot_10:.2f} s")
if "cfg5" in which:
    N, C, n = 32768, 1, 32768
    zs = [(torch.randn(N, 4096, device=dev), torch.randn(n, 4096, device=dev) / N**0.5),
          (torch.randn(N, 4096, device=dev), torch.randn(n, 1000, device=dev) / N**0.5)]
    G = torch.empty(n, n, device=dev)

    def step():
        for i, (z, s) in enumerate(zs):
            Gz = kernels.gram_syrk(z)
            Gs = kernels.gram_syrk(s)
            kernels.gram_hadamard(Gz, Gs, C, N, out=G, alpha=1.0, beta=0.0 if i == 0 else 1.0)
        return kernels.symeig(G, eigenvectors=True, overwrite=False)

    t, _ = timed(step, reps=1)

    def step10():
        for i, (z, s) in enumerate(zs):
            Gz = kernels.gram_syrk(z)
            Gs = kernels.gram_syrk(s)
            kernels.gram_hadamard(Gz, Gs, C, N, out=G, alpha=1.0, beta=0.0 if i == 0 else 1.0)
        plan = kernels.symeig_reduce(G, overwrite=False)
        return plan.evals, plan.select(list(range(n - 10, n)))

    t10, _ = timed(step10, reps=1)
    print(f"cfg5 n=32768 factorised Gram (2 Linear layers) + symeig(vectors): {t:.2f} s -> {n/t:.0f} eigenpairs/s; top-10: {t10:.2f} s")
