"""Time of the Gram SYRK's split pass and product on the headline shape's first chunks (kernel names from the profiler are
not needed: the SYRK of a [40960, 8192] slice is two chunks = two split launches + two products)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
A = torch.randn(40960, 8192, device="cuda") * 0.01
G = torch.empty(40960, 40960, device="cuda")
for _ in range(2): kernels.gram_syrk(A, out=G)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): kernels.gram_syrk(A, out=G)
e1.record(); torch.cuda.synchronize()
print(f"SYRK 40960 x 8192: {e0.elapsed_time(e1) / 5:.2f} ms per call")
