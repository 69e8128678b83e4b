"""Quick throughput probe of the MFMA GEMM / SYRK kernels (run on the GPU box)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vivit_amd import kernels


def timeit(fn, iters=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), flush=True)
for n, p in [(4096, 4096), (8192, 8192), (8192, 65536), (16384, 16384), (1280, 401408)]:
    A = torch.randn(n, p, device=dev)
    G = torch.empty(n, n, device=dev)
    t = timeit(lambda: kernels.gram_syrk(A, out=G))
    print(f"syrk n={n} p={p}: {t*1e3:.2f} ms  {n*(n+1)*p/t/1e12:.1f} TFLOP/s (sym flops)", flush=True)
    if n <= 8192 and p <= 8192:
        B = torch.randn(n, p, device=dev)
        t = timeit(lambda: kernels.gemm_nt(A, B, out=G))
        print(f"gemm_nt {n}x{n}x{p}: {t*1e3:.2f} ms  {2*n*n*p/t/1e12:.1f} TFLOP/s", flush=True)
        Bt = B.T.contiguous()
        t = timeit(lambda: kernels.gemm_nn(A, Bt, out=G))
        print(f"gemm_nn {n}x{n}x{p}: {t*1e3:.2f} ms  {2*n*n*p/t/1e12:.1f} TFLOP/s", flush=True)
        t = timeit(lambda: torch.matmul(A, B.T))
        print(f"torch(hipblas) {n}x{n}x{p}: {t*1e3:.2f} ms  {2*n*n*p/t/1e12:.1f} TFLOP/s", flush=True)
    del A, G
