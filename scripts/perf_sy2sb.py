import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = int(sys.argv[1])
M = torch.randn(n, n, device=dev)
S = (M + M.T) / 2
del M
kernels.sy2sb(S)
torch.cuda.synchronize()
t0 = time.perf_counter()
kernels.sy2sb(S)
torch.cuda.synchronize()
print(f"sy2sb n={n}: {(time.perf_counter()-t0)*1e3:.1f} ms")
