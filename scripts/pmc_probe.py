import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vivit_amd import kernels
what = sys.argv[1]
dev = torch.device("cuda:0")
if what == "syrk":
    A = torch.randn(8192, 16384, device=dev)
    G = kernels.gram_syrk(A)
    torch.cuda.synchronize()
    print("syrk done", float(G[0, 0]))
else:
    n = int(sys.argv[2])
    M = torch.randn(n, n, device=dev)
    S = (M + M.T) / 2
    d, e, tau, A = kernels.sytrd(S)
    torch.cuda.synchronize()
    print("sytrd done", float(d[0]))
