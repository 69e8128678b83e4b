"""Deep-K skinny products of the band reduction (64 x N x K, both operands K-contiguous): time per call."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
K = 40960
for N in (64, 128, 384):
    A = torch.randn(64, K, device="cuda"); B = torch.randn(N, K, device="cuda")
    out = torch.empty(64, N, device="cuda")
    for _ in range(3): kernels.gemm_nt(A, B, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): kernels.gemm_nt(A, B, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    err = (out.double() - A.double() @ B.double().T).abs().max().item()
    print(os.environ.get("TAG", ""), f"N={N}: {us:.1f} us per call ({(64 + N) * K * 4 / us / 1e6:.2f} TB/s of operand bytes), max err {err:.2e}", flush=True)
