"""Timing probe of the eigensolver stages on the GPU box."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vivit_amd import kernels

sizes = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1280, 4096, 8192]
dev = torch.device("cuda:0")
for n in sizes:
    g = torch.Generator(device=dev).manual_seed(n)
    V = torch.randn(n, 2 * n, device=dev, generator=g) * (0.999 ** torch.arange(2 * n, device=dev))
    S = kernels.gram_syrk(V)
    del V
    for vec in [False, True]:
        kernels.symeig(S, eigenvectors=vec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w, Z = kernels.symeig(S, eigenvectors=vec)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f"symeig n={n} vectors={vec}: {t*1e3:.1f} ms   ({n/t:.0f} eigenpairs/s)  tridiag-bytes-roofline {(2/3)*n**3/t/1e12:.2f} TB/s-equiv", flush=True)
    d, e, tau, A = kernels.sytrd(S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d, e, tau, A = kernels.sytrd(S)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"  sytrd n={n}: {t*1e3:.1f} ms  -> {(2/3)*n**3/t/1e12:.3f} TB/s algorithmic", flush=True)
    t0 = time.perf_counter()
    w, Z = kernels.stedc(d, e, eigenvectors=True)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"  stedc(D&C) n={n}: {t*1e3:.1f} ms", flush=True)
    t0 = time.perf_counter()
    w, _ = kernels.stedc(d, e, eigenvectors=False)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"  stebz(bisection) n={n}: {t*1e3:.1f} ms", flush=True)
    if n <= 8192:
        t0 = time.perf_counter()
        torch.linalg.eigh(S)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f"  torch.linalg.eigh (hipSOLVER) n={n}: {t*1e3:.1f} ms", flush=True)
    del S, Z, A
