import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = int(sys.argv[1])
M = torch.randn(n, 2*n, device=dev) / (2*n)**0.5
S = M @ M.T
for vec in [True, False]:
    kernels.symeig(S, eigenvectors=vec); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): kernels.symeig(S, eigenvectors=vec)
    torch.cuda.synchronize()
    print(f"symeig n={n} vectors={vec}: {(time.perf_counter()-t0)/5*1e3:.1f} ms")
