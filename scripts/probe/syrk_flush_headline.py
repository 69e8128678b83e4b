"""Headline-shape Gram SYRK (n = 40 960, K = 401 408) for one setting of the chain length (env VIVIT_BX_FLUSH*): time of
3 runs + diagonal bias + sampled off-diagonal error."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ctypes
import vivit_amd._lib as L
if os.environ.get("VIVIT_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["VIVIT_LIB"])
    _probe = ctypes.CDLL(L.LIB_PATH)
    _probe.vivit_hip_abi_version.restype = ctypes.c_int
    L.ABI_VERSION = _probe.vivit_hip_abi_version()
    L.SIGNATURES = {k: v for k, v in L.SIGNATURES.items() if hasattr(_probe, k)}
from vivit_amd import kernels
dev = torch.device("cuda:0")
n, p = 40960, int(os.environ.get("P", 401408))
pad = int(os.environ.get('PAD', '0'))
G = torch.empty(n, n + pad, device=dev)[:, :n]
A = torch.randn(n, p, device=dev)
kernels.gram_syrk(A, out=G); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    kernels.gram_syrk(A, out=G); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
dt = min(ts)
ri = torch.arange(5, n, 997, device=dev)
ref = A[ri].double() @ A[ri].double().T
got = G[ri][:, ri].double()
off = ~torch.eye(len(ri), dtype=torch.bool, device=dev)
unit = p ** 0.5
print(f"pad={pad} flush={os.environ.get('VIVIT_BX_FLUSH')} diag={os.environ.get('VIVIT_BX_FLUSH_DIAG')}: {dt*1e3:.1f} ms (all {[round(t*1e3,1) for t in ts]})"
      f" {n*(n+1)*p/dt/1e12:.1f} TFLOP/s  off_rms {((got-ref)[off]/unit).pow(2).mean().sqrt().item():.2e}"
      f" diag_mean {((got-ref).diagonal()/ref.diagonal()).mean().item():+.2e}", flush=True)
