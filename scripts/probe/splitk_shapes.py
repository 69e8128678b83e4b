"""Small-output / deep-K products of the bf16-pipe tile kernel (split-K form): time and TFLOP/s of the Gram SYRK at the shapes
BASELINE config 1 and the back-transformation's S = Y Y^T use.   python scripts/probe/splitk_shapes.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
for n, K in ((1280, 407050), (1280, 401408), (2048, 40960), (2560, 407050), (5120, 401408), (1024, 464154 // 16 * 16)):
    A = torch.randn(n, K, device=dev)
    G = kernels.gram_syrk(A)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); kernels.gram_syrk(A, out=G); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    print(f"n = {n:5d}  K = {K:7d}: {t * 1e3:7.3f} ms = {n * (n + 1) * K / t / 1e12:6.1f} TF", flush=True)
# a product with several accumulation chains per launch tile (the shape of Q1's W1 = Zt Y^T): chain ends inside the K loop
m, nn, k = 40960, 2048, 20480
A = torch.randn(m, k, device=dev); B = torch.randn(nn, k, device=dev); C = torch.empty(m, nn, device=dev)
kernels.gemm_nt(A, B, out=C, alpha=1.0, beta=0.0); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); kernels.gemm_nt(A, B, out=C, alpha=1.0, beta=0.0); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = sorted(ts)[2]
print(f"gemm_nt {m} x {nn} x {k}: {t * 1e3:7.3f} ms = {2 * m * nn * k / t / 1e12:6.1f} TF", flush=True)
