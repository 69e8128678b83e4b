"""fp32 GEMM on the bf16 pipe (VIVIT_GEMM_SPLIT=3|6|9) against the fp32-MFMA kernel (0): accuracy vs fp64, speed."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
mode = os.environ.get("VIVIT_GEMM_SPLIT", "0")
g = torch.Generator(device=dev).manual_seed(0)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
# accuracy: SYRK n = 4096, K = 65536 (8 flush chunks), entries vs fp64
n, K = 4096, 65536
A = torch.randn(n, K, device=dev, generator=g)
G = kernels.gram_syrk(A)
I = torch.arange(0, n, 37, device=dev)
ref = A[I].double() @ A.double().T
sq = (A.double() ** 2).sum(1)
err = ((G[I].double() - ref).abs() / torch.sqrt(sq[I][:, None] * sq[None, :])).max().item()
derr = ((G.diagonal().double() - sq).abs() / sq).max().item()
print(f"mode {mode}: SYRK n={n} K={K} entry_err {err:.2e} diag_err {derr:.2e} symmetric {torch.equal(G, G.T)}", flush=True)
# short-K accuracy (per-product error visible): K = 1024
A2 = torch.randn(2048, 1024, device=dev, generator=g); B2 = torch.randn(2048, 1024, device=dev, generator=g)
C2 = kernels.gemm_nt(A2, B2)
ref2 = A2.double() @ B2.double().T
print(f"mode {mode}: NT 2048x2048x1024 max err / (|a||b|) {((C2.double()-ref2).abs().max() / (A2.double().norm(dim=1).max()*B2.double().norm(dim=1).max())).item():.2e}", flush=True)
del A, G, ref
# speed
m = 16384
A = torch.randn(m, m, device=dev, generator=g); B = torch.randn(m, m, device=dev, generator=g)
dt = t(lambda: kernels.gemm_nt(A, B))
print(f"mode {mode}: NT {m}^3 {dt*1e3:.1f} ms = {2*m**3/dt/1e12:.1f} TFLOP/s-equivalent", flush=True)
dt = t(lambda: kernels.gram_syrk(A))
print(f"mode {mode}: SYRK {m}x{m} {dt*1e3:.1f} ms = {m*(m+1)*m/dt/1e12:.1f} TFLOP/s-equivalent", flush=True)
if os.environ.get("HEADLINE"):
    del A, B
    A = torch.randn(40960, 401408, device=dev); G = torch.empty(40960, 40960, device=dev)
    dt = t(lambda: kernels.gram_syrk(A, out=G), reps=1)
    print(f"mode {mode}: headline SYRK {dt:.3f} s = {40960*40961*401408/dt/1e12:.1f} TFLOP/s-equivalent", flush=True)
