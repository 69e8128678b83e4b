import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivit_amd import kernels
dev = torch.device("cuda:0")
for n in [int(x) for x in sys.argv[1].split(",")]:
    V = torch.randn(n, 2 * n, device=dev) * (0.9995 ** torch.arange(2 * n, device=dev))
    S = kernels.gram_syrk(V)
    del V
    kernels.symeig(S, eigenvectors=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w, Z = kernels.symeig(S, eigenvectors=True)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"symeig(vectors) n={n} two_stage={os.environ.get('VIVIT_TWO_STAGE','auto')}: {t*1e3:.1f} ms")
    del S, Z
