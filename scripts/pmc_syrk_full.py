"""The headline Gram SYRK (n = 40960, P = 401408) once, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
A = torch.randn(40960, 401408, device=dev)
G = torch.empty(40960, 40960, device=dev)
kernels.gram_syrk(A, out=G)
torch.cuda.synchronize()
print("done", float(G[0, 0]))
