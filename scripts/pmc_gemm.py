"""One big fp32 GEMM (and the torch/hipBLASLt one for comparison) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vivit_amd import kernels

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = torch.randn(n, n, device=dev)
B = torch.randn(n, n, device=dev)
C = torch.empty(n, n, device=dev)
for _ in range(2):
    kernels.gemm_nt(A, B, out=C)
    torch.matmul(A, B.T, out=C)
torch.cuda.synchronize()
