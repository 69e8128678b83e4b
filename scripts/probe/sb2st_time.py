"""Wall time of the bulge chase alone at n (default 40960) with the library of VIVIT_LIB (timing-only variants of the persistent
kernel: -DSB2ST_PVAR=1 no arithmetic, =2 no waiting)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import _lib
if os.environ.get("VIVIT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VIVIT_LIB"])
from vivit_amd import kernels
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
AB = torch.randn(n, 129, device="cuda")
AB[:, :64] = 0
for rep in range(3):
    A = AB.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    kernels.sb2st(A)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(os.environ.get("VIVIT_LIB", "product"), f"n={n}: sb2st {dt * 1e3:.1f} ms (incl. allocation of R2)", flush=True)
