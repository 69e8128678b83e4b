import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = 8192
B = torch.randn(n, n, device=dev)
for m in [64, 128, 256]:
    A = torch.randn(m, n, device=dev)
    for name, fn in [("nt", lambda: kernels.gemm_nt(A, B)), ("torch", lambda: A @ B.T)]:
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name} {m}x{n}x{n}: {dt*1e3:.3f} ms  {2*m*n*n/dt/1e12:.1f} TFLOP/s")
