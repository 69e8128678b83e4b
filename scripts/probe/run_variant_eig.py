"""Timing-only experiment: two-stage symeig with an experimental build of the library (results invalid)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vivit_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = 40960
V = torch.randn(n, 2 * n, device=dev) / (2 * n) ** 0.5
S = kernels.gram_syrk(V); del V
lib = L.load()
for it in range(2):
    A = S.clone(); w = torch.empty(n, device=dev); Z = torch.empty(n, n, device=dev); info = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib.vivit_symeig_f32_workspace_bytes(n, 1); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lib.vivit_symeig_f32(A.data_ptr(), n, n, w.data_ptr(), Z.data_ptr(), n, ws.data_ptr(), wsb, info.data_ptr(), None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(os.path.basename(sys.argv[1]), f"{dt*1e3:.0f} ms")
