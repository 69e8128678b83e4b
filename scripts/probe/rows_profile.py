import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = 40960
V = torch.randn(n, 2 * n, device=dev) / (2 * n) ** 0.5
S = kernels.gram_syrk(V); del V
for _ in range(2):
    w, Zt = kernels.symeig_rows(S, 35840, 40960)
torch.cuda.synchronize()
