"""One launch each of the two MFMA-bound kernels the roofline table names, for a `rocprofv3 --pmc` pass
(SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE; restrict the counters to these kernels with
--kernel-include-regex "gemm256_kernel|q2_apply_kernel"):
  * the headline Gram SYRK (n = 40960, P = 401408) -- gemm256_kernel<0, 0>
  * the Q2 back-transformation inside a two-stage symeig with vectors at n = 8192 (q2_apply_kernel; the eigensolver at
    n = 40960 issues 8e4 bulge-chasing dispatches, too many for a counter run)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vivit_amd import kernels

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("syrk", "both"):
    A = torch.randn(40960, 401408, device=dev)
    G = torch.empty(40960, 40960, device=dev)
    kernels.gram_syrk(A, out=G)
    torch.cuda.synchronize()
    print("syrk done", float(G[0, 0]), flush=True)
    del A, G
if which in ("q2", "both"):
    n = 8192
    V = torch.randn(n, n // 2, device=dev)
    G = kernels.gram_syrk(V)
    w, Z = kernels.symeig(G, eigenvectors=True, overwrite=True)
    torch.cuda.synchronize()
    print("symeig done", float(w[-1]), flush=True)
