import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vivit_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = 40960
M = torch.randn(n, n, device=dev); S = (M + M.T) / 2; del M
kernels.sy2sb(S); torch.cuda.synchronize()
t0 = time.perf_counter(); kernels.sy2sb(S); torch.cuda.synchronize()
print(os.path.basename(sys.argv[1]), f"{(time.perf_counter()-t0)*1e3:.0f} ms")
