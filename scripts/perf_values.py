import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivit_amd import kernels
dev = torch.device("cuda:0")
for n in [int(x) for x in sys.argv[1].split(",")]:
    M = torch.randn(n, n, device=dev)
    S = (M + M.T) / 2
    del M
    kernels.symeig(S, eigenvectors=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w, _ = kernels.symeig(S, eigenvectors=False)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"eigvalsh n={n} two_stage={os.environ.get('VIVIT_TWO_STAGE','auto')}: {t*1e3:.1f} ms")
    AB, tau1, A = kernels.sy2sb(S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    AB, tau1, A = kernels.sy2sb(S)
    torch.cuda.synchronize()
    print(f"   sy2sb n={n}: {(time.perf_counter()-t0)*1e3:.1f} ms")
    del S, AB, A
