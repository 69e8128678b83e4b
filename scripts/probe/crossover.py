"""symeig with vectors / values only at one size, one-stage vs two-stage (VIVIT_TWO_STAGE is read once per process)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
n = int(sys.argv[1]); vec = sys.argv[2] == "1"
V = torch.randn(n, n // 2, device="cuda")
G = kernels.gram_syrk(V); del V
kernels.symeig(G, eigenvectors=vec); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): kernels.symeig(G, eigenvectors=vec)
torch.cuda.synchronize()
print(f"n={n} vectors={int(vec)} two_stage={os.environ.get('VIVIT_TWO_STAGE', 'auto')}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms", flush=True)
