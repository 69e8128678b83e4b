"""symeig (vectors) at n = 40960 once on a Gram matrix, printing the library's stage marks (ms)."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import _lib
if os.environ.get("VIVIT_LIB"):   # A/B against another build of the library (e.g. last round's: scripts/probe/libr02.so)
    _lib.LIB_PATH = os.path.abspath(os.environ["VIVIT_LIB"])
    _probe = ctypes.CDLL(_lib.LIB_PATH)
    _probe.vivit_hip_abi_version.restype = ctypes.c_int
    _lib.ABI_VERSION = _probe.vivit_hip_abi_version()
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_probe, k)}
from vivit_amd import kernels
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
V = torch.randn(n, n // 2, device="cuda")
pad = int(os.environ.get("PAD", "0"))
G = torch.empty(n, n + pad, device="cuda")[:, :n]
kernels.gram_syrk(V, out=G)
lib = _lib.load()
kernels.symeig(G, eigenvectors=True, overwrite=True)   # (in place: a padded leading dimension survives)
kernels.gram_syrk(V, out=G); del V
torch.cuda.synchronize()
lib.vivit_profile_begin(64)
w, Z = kernels.symeig(G, eigenvectors=True, overwrite=True)
torch.cuda.synchronize()
ms = (ctypes.c_double * 16)()
lib.vivit_profile_stages(ms, 16)
lib.vivit_profile_end((ctypes.c_double * 6)())
print(os.environ.get("TAG", ""), " ".join("%.0f" % x for x in ms), "total %.0f ms" % sum(ms), flush=True)
