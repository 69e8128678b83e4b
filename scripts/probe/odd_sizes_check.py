"""One-off robustness check of the full eigensolver at sizes no test uses (odd, not multiples of the tile sizes): residual and
orthonormality in fp64 accumulation on sampled columns, trace identity."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [12289, 16100, 20001]:
    g = torch.Generator(device=dev).manual_seed(n)
    V = torch.randn(n, n // 3 + 7, device=dev, generator=g)
    G = kernels.gram_syrk(V) / V.shape[1]
    G0 = G.clone()
    w, Z = kernels.symeig(G, eigenvectors=True, overwrite=True)
    idx = torch.randperm(n, device=dev)[:64]
    Zs = Z[:, idx].double()
    R = G0.double() @ Zs - Zs * w[idx].double()
    res = (R.norm(dim=0).max() / w.abs().max().double()).item()
    orth = ((Z.double().T @ Zs) - torch.eye(n, device=dev, dtype=torch.float64)[:, idx]).abs().max().item()
    tr = abs(w.double().sum().item() - G0.double().diagonal().sum().item()) / G0.double().diagonal().sum().item()
    print(f"n={n}: residual {res:.2e} orthonormality {orth:.2e} trace {tr:.2e} ascending {bool((w[1:] >= w[:-1]).all())}", flush=True)
    del V, G, G0, Z, Zs, R
    torch.cuda.empty_cache()
