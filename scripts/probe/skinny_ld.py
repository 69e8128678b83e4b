"""Channel-camping probe: the 64-row streaming GEMM with B's leading dimension a multiple of 4 KB vs padded."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
for n in [8192, 40960]:
    A = torch.randn(64, n, device=dev)
    for pad in [0, 16, 64, 272]:
        Bfull = torch.randn(n, n + pad, device=dev)
        B = Bfull[:, :n]
        for name, fn in [("nt", lambda: kernels.gemm_nt(A, B)), ("nn", lambda: kernels.gemm_nn(A, B))]:
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            print(f"{name} n={n} ld=n+{pad}: {dt*1e3:.3f} ms  {4*n*n/dt/1e12:.2f} TB/s")
        del Bfull, B
