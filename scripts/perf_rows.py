"""Per-rank unit of the multi-GPU eigensolver at full size: symeig_rows on 1/R of the eigenvectors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = int(sys.argv[1]); R = int(sys.argv[2])
V = torch.randn(n, 2 * n, device=dev) / (2 * n) ** 0.5
S = kernels.gram_syrk(V); del V
per = -(-n // R)
for r in [0, R - 1]:
    lo, hi = r * per, min((r + 1) * per, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w, Zt = kernels.symeig_rows(S, lo, hi)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    resid = (Zt @ S - w[lo:hi, None] * Zt).abs().max().item() / w.abs().max().item()
    orth = (Zt @ Zt.T - torch.eye(hi - lo, device=dev)).abs().max().item()
    print(f"n={n} rank {r}/{R} rows [{lo},{hi}): {dt*1e3:.1f} ms  resid {resid:.2e} orth {orth:.2e}")
