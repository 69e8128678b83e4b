"""Deep-K skinny product 64 x N x K with padded leading dimensions (row stride K + pad floats): channel-conflict probe."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
K = 40960
for pad in (0, 64, 256, 1024, 4160):
    for N in (64, 384):
        Ab = torch.randn(64, K + pad, device="cuda"); Bb = torch.randn(N, K + pad, device="cuda")
        A, B = Ab[:, :K], Bb[:, :K]
        out = torch.empty(64, N, device="cuda")
        for _ in range(3): kernels.gemm_nt(A, B, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): kernels.gemm_nt(A, B, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"pad={pad} N={N}: {us:.1f} us per call ({(64 + N) * K * 4 / us / 1e6:.2f} TB/s)", flush=True)
