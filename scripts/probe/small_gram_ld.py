"""Same product with the operands' leading dimension n (= 40 x 4 KB) vs n + pad: L2 channel camping probe."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = 40960
for pad in [0, 16, 32, 64, 128, 256, 1040]:
    X = torch.randn(64, n + pad, device=dev); Y = torch.randn(64, n + pad, device=dev)
    a, b = X[:, :40000], Y[:, :40000]
    kernels.gemm_nt(a, b); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): kernels.gemm_nt(a, b)
    torch.cuda.synchronize()
    print(f"ld = n + {pad}: {(time.perf_counter()-t0)/20*1e6:.1f} us")
