"""Timing-only experiment: run the fp32 GEMM with an experimental build of the library (results invalid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vivit_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
m = n = k = 16384
A = torch.randn(m, k, device=dev); B = torch.randn(n, k, device=dev)
kernels.gemm_nt(A, B); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): kernels.gemm_nt(A, B)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(os.path.basename(sys.argv[1]), f"{dt*1e3:.2f} ms {2*m*n*k/dt/1e12:.1f} TFLOP/s")
