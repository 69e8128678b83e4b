"""The band reduction's panel product P^T = V^T A22 (64 x m x m, gemm64_dma_kernel): time per call for a library build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vivit_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from vivit_amd import kernels
for m in (40960, 20480):
    A = torch.randn(64, m, device="cuda"); B = torch.randn(m, m, device="cuda")
    out = torch.empty(64, m, device="cuda")
    for _ in range(2): kernels.gemm_nn(A, B, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): kernels.gemm_nn(A, B, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ref = A[:, :4096].double() @ B[:4096, :256].double()
    print(os.path.basename(sys.argv[1]), f"m={m}: {ms:.3f} ms ({m * m * 4 / ms / 1e9:.2f} TB/s, {2 * 64 * m * m / ms / 1e9:.0f} TFLOP/s)", flush=True)
    del A, B, out
