import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivit_amd import kernels
dev = torch.device("cuda:0")
n, P = 16384, 262144
V = torch.randn(n, P, device=dev)
for K in [1, 2, 8, 16]:
    E = torch.randn(K, n, device=dev)
    kernels.gemm_nn(E, V)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        kernels.gemm_nn(E, V)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    print(f"backproject K={K} n={n} P={P}: {t*1e3:.2f} ms  {4*n*P/t/1e12:.2f} TB/s (V streamed once), {2*K*n*P/t/1e12:.1f} TFLOP/s")
