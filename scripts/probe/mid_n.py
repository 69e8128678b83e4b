"""eigvalsh / symeig wall time at mid sizes (193 < n <= 2048), persistent one-XCD tridiagonalisation on and off
(VIVIT_SYTRD_PERSIST is read once per process: run twice)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels

def med(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3

for n in (256, 512, 768, 1024, 1280, 1536, 2048):
    V = torch.randn(n, 2 * n, device="cuda")
    G = V @ V.T
    w64 = torch.linalg.eigvalsh(G.double().cpu())
    w = kernels.symeig(G.clone(), eigenvectors=False, overwrite=True)[0]
    err = float((w.double().cpu() - w64).abs().max() / w64.abs().max())
    t_vals = med(lambda: kernels.symeig(G.clone(), eigenvectors=False, overwrite=True))
    t_vecs = med(lambda: kernels.symeig(G.clone(), eigenvectors=True, overwrite=True))
    print(f"n={n}: eigvalsh {t_vals:.2f} ms, symeig {t_vecs:.2f} ms, max rel err {err:.2e} (persist={os.environ.get('VIVIT_SYTRD_PERSIST', '1')})", flush=True)
