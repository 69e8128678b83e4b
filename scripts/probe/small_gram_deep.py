"""Gram matrices of small batches with a deep contraction (n = 1024 ... 2048, P = 4e5): ms per SYRK."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
for n, p in ((1280, 401408), (1024, 470000), (2048, 401408), (512, 401408)):
    V = torch.randn(n, p, device="cuda") / 30
    for _ in range(2): G = kernels.gram_syrk(V)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): G = kernels.gram_syrk(V)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ref = V[:64].double() @ V[:256].double().T
    err = ((G[:64, :256].double() - ref).abs().max() / ref.abs().max()).item()
    print(os.environ.get("TAG", ""), f"n={n} P={p}: {ms:.2f} ms ({n * (n + 1) * p / ms / 1e9:.0f} TFLOP/s), rel err {err:.1e}, symmetric {bool(torch.equal(G, G.T))}", flush=True)
    del V, G
