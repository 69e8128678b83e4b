"""One size of the band reduction's panel product (64 x m x m) a few times: the workload of a counter pass."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
m = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
A = torch.randn(64, m, device="cuda"); B = torch.randn(m, m, device="cuda")
out = torch.empty(64, m, device="cuda")
for _ in range(4): kernels.gemm_nn(A, B, out=out)
torch.cuda.synchronize()
