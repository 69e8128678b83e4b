"""Timing of the GEMM shapes the eigensolver issues, against any build of the library (VIVIT_LIB)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from vivit_amd import _lib
if os.environ.get("VIVIT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VIVIT_LIB"])
    _probe = ctypes.CDLL(_lib.LIB_PATH)
    _probe.vivit_hip_abi_version.restype = ctypes.c_int
    _lib.ABI_VERSION = _probe.vivit_hip_abi_version()
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(_probe, k)}
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
tag = os.environ.get("TAG", "")

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

n = 32768
Z = torch.randn(n, n, device=dev)
Y = torch.randn(2048, n, device=dev) / n ** 0.5      # reflector block, k-major rows
W = torch.randn(n, 2048, device=dev)
out1 = torch.empty(n, 2048, device=dev)
t_a = timed(lambda: kernels.gemm_nt(Z, Y, out=out1))                       # Zt Y^T-type: K = n, narrow output
t_c = timed(lambda: kernels.gemm_nn(W, Y, out=Z, alpha=-1.0, beta=1.0), 3)  # Zt -= W Y: K = 2048, n x n output, beta = 1
T = torch.randn(2048, 2048, device=dev)
t_b = timed(lambda: kernels.gemm_nt(out1, T, out=W))                       # (Zt Y) T^T
A = torch.randn(n, 512, device=dev)
G = torch.randn(n, n, device=dev); G = G + G.T
t_u = timed(lambda: kernels.gram_syrk(A, out=G, alpha=-1.0, beta=1.0), 3)   # rank-512 symmetric update, mirrored
print(f"{tag}: Zt*Y (K=n) {t_a:.2f} ms | *T {t_b:.2f} ms | Zt -= W Y (K=2048, beta=1) {t_c:.2f} ms | rank-512 SYRK update {t_u:.2f} ms", flush=True)
