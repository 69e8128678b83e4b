"""The headline Gram SYRK (n = 40960, P = 401408) once, for rocprofv3 --pmc passes.  Data: N(0,1) (default) or, with the argument `bench`,
the benchmark's own first-layer factor (ReLU-masked: about half of its 784-column groups are zero, bench.mlp_sqrt_ggn_factors)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "bench":
    import bench
    A = bench.mlp_sqrt_ggn_factors((784, 512, 10), 4096, dev)[2]
else:
    A = torch.randn(40960, 401408, device=dev)
G = torch.empty(40960, 40960, device=dev)
kernels.gram_syrk(A, out=G)
torch.cuda.synchronize()
print("done", tuple(A.shape), float(G[0, 0]))
