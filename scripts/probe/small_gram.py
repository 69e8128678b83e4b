"""C[64,64] = X[64,m] Y[64,m]^T (k-major operands with ld = n): the S / S2 / G12 products of the band reduction."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = 40960
X = torch.randn(64, n, device=dev); Y = torch.randn(128, n, device=dev)
for m in [40000, 20000, 5000]:
    for rows in [64, 128]:
        a, b = X[:, :m], Y[:rows, :m]
        kernels.gemm_nt(a, b); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): kernels.gemm_nt(a, b)
        torch.cuda.synchronize()
        print(f"m={m} out 64x{rows}: {(time.perf_counter()-t0)/20*1e6:.1f} us")
