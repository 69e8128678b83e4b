"""The band reduction's streaming panel product P^T = V^T A22 (64 x m x m on gemm64_bx_kernel) by trailing size m: time, TB/s and the
launch geometry the host picked.   python scripts/probe/panel_product_sizes.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
tot_t = tot_b = 0.0
for m in (40896, 36864, 32768, 28672, 24576, 20480, 16384, 12288, 8192, 4096, 2048):
    A = torch.randn(64, m, device="cuda"); B = torch.randn(m, m, device="cuda")
    out = torch.empty(64, m, device="cuda")
    for _ in range(2): kernels.gemm_nn(A, B, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): kernels.gemm_nn(A, B, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"m={m:6d}: {ms * 1e3:8.1f} us  {m * m * 4 / ms / 1e9:5.2f} TB/s", flush=True)
    tot_t += ms * 64 * (4096 / 64 if m > 2048 else 32); tot_b += m * m * 4
    del A, B, out
