"""The band reduction's panel product with a padded leading dimension of the streamed matrix (HBM channel-mapping probe)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
m = 40960
A = torch.randn(64, m, device="cuda")
for pad in (0, 64, 256, 1024, 4160):
    Bb = torch.randn(m, m + pad, device="cuda"); B = Bb[:, :m]
    out = torch.empty(64, m, device="cuda")
    for _ in range(2): kernels.gemm_nn(A, B, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): kernels.gemm_nn(A, B, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"pad={pad}: {ms:.3f} ms ({m * m * 4 / ms / 1e9:.2f} TB/s)", flush=True)
    del Bb, B, out
