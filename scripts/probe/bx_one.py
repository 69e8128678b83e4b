"""One 16384^3 NT GEMM (for rocprofv3 --pmc passes on the split-bf16 kernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
m = 16384
A = torch.randn(m, m, device=dev); B = torch.randn(m, m, device=dev)
kernels.gemm_nt(A, B); kernels.gemm_nt(A, B)
torch.cuda.synchronize()
print("done")
