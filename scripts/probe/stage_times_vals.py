"""eigvalsh (values only) at n once on a Gram matrix, printing the library's stage marks (ms)."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import _lib, kernels
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
V = torch.randn(n, n // 2, device="cuda")
G = torch.empty(n, n, device="cuda")
kernels.gram_syrk(V, out=G)
lib = _lib.load()
kernels.symeig(G.clone(), eigenvectors=False, overwrite=True)
torch.cuda.synchronize()
lib.vivit_profile_begin(64)
w, _ = kernels.symeig(G, eigenvectors=False, overwrite=True)
torch.cuda.synchronize()
ms = (ctypes.c_double * 16)()
lib.vivit_profile_stages(ms, 16)
lib.vivit_profile_end((ctypes.c_double * 6)())
print(f"n={n} eigvalsh stages (begin prep sy2sb sb2st tridiag ...):", " ".join("%.1f" % x for x in ms), "total %.1f ms" % sum(ms), flush=True)
