"""Is the 5 s Gram SYRK clock-limited?  Same n, increasing P (run time), TFLOP/s each."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = 40960
G = torch.empty(n, n, device=dev)
for p in [16384, 65536, 131072, 401408]:
    A = torch.randn(n, p, device=dev)
    kernels.gram_syrk(A, out=G); torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernels.gram_syrk(A, out=G); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"syrk n={n} p={p}: {dt*1e3:.1f} ms  {n*(n+1)*p/dt/1e12:.1f} TFLOP/s")
    del A
