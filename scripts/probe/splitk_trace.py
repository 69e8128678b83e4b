"""One small-output Gram SYRK (split-K form) for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 scripts/probe/splitk_trace.py [n K]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
n, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 401408)
A = torch.randn(n, K, device="cuda:0")
G = kernels.gram_syrk(A)
for _ in range(4):
    kernels.gram_syrk(A, out=G)
torch.cuda.synchronize()
