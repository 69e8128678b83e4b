import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
for n, p in [(1280, 401408), (1024, 470016), (2560, 100000)]:
    A = torch.randn(n, p, device=dev)
    G = torch.empty(n, n, device=dev)
    kernels.gram_syrk(A, out=G); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): kernels.gram_syrk(A, out=G)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"syrk n={n} p={p}: {dt*1e3:.2f} ms {n*(n+1)*p/dt/1e12:.1f} TF")
