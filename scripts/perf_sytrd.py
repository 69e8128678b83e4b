import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivit_amd import kernels
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda:0")
M = torch.randn(n, n, device=dev)
S = (M + M.T) / 2
del M
kernels.sytrd(S)
torch.cuda.synchronize()
t0 = time.perf_counter()
kernels.sytrd(S)
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"sytrd n={n}: {t*1e3:.1f} ms -> {(2/3)*n**3/t/1e12:.3f} TB/s algorithmic, {t/n*1e6:.1f} us/column")
