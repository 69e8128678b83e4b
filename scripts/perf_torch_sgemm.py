"""Reference point for the fp32 GEMM rate on this GPU: torch.matmul (rocBLAS/hipBLASLt) vs vivit_gemm_nt."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vivit_amd import kernels

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
for (m, n, k) in [(8192, 8192, 8192), (16384, 16384, 16384), (40960, 4096, 65536)]:
    A = torch.randn(m, k, device=dev)
    B = torch.randn(n, k, device=dev)
    for name, fn in [("torch", lambda: A @ B.T), ("vivit_nt", lambda: kernels.gemm_nt(A, B))]:
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{name} {m}x{n}x{k}: {dt*1e3:.2f} ms  {2*m*n*k/dt/1e12:.1f} TFLOP/s")
