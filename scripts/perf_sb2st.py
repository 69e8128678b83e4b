import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivit_amd import kernels
dev = torch.device("cuda:0")
for n in [8192, 40960]:
    AB = torch.randn(n, 129, device=dev)
    AB[:, :64] = 0
    kernels.sb2st(AB[:2048].contiguous())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d, e, R2, tau2 = kernels.sb2st(AB)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"sb2st n={n}: {t*1e3:.1f} ms ({t/(2*n)*1e6:.2f} us per wavefront step)")
    del R2
