"""The panel product of the band reduction in isolation: [64, K] x [N, K]^T with N = K (streams B once)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
for n in [8192, 20480, 40960]:
    A = torch.randn(64, n, device=dev)
    B = torch.randn(n, n, device=dev)
    for name, fn in [("nt", lambda: kernels.gemm_nt(A, B)), ("nn", lambda: kernels.gemm_nn(A, B)), ("torch", lambda: A @ B.T)]:
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name} 64x{n}x{n}: {dt*1e3:.3f} ms  {2*64*n*n/dt/1e12:.1f} TFLOP/s  {4*n*n/dt/1e12:.2f} TB/s")
