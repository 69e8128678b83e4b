"""Host-side cost of one launch through the ctypes binding (tiny tensors: the GPU work is nothing): wall time per call of a
few wrappers, with the queue drained first and the calls issued back to back."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
M = torch.randn(1, 8, 4, 6, 6, device=dev); x = torch.randn(8, 4, 6, 6, device=dev)
w = torch.randn(4, 4, 3, 3, device=dev); sc = torch.rand(4, device=dev)
cases = {
    "act_jac_t": lambda: kernels.act_jac_t(M, x, "relu"),
    "channel_scale": lambda: kernels.channel_scale(M, sc),
    "row_dot": lambda: kernels.row_dot(M.reshape(-1, 36)),
    "conv2d_jac_t": lambda: kernels.conv2d_jac_t(M, w, (6, 6), (1, 1), (1, 1), (1, 1)),
    "conv2d_weight_mjp": lambda: kernels.conv2d_weight_mjp(M, x, (3, 3), (1, 1), (1, 1), (1, 1)),
    "torch.mul (reference)": lambda: torch.mul(M, 2.0),
}
for name, fn in cases.items():
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:24s} {(t1 - t0) / 2000 * 1e6:6.1f} us per call (host)", flush=True)
