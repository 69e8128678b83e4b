"""symeig / eigvalsh wall time at n around the one-/two-stage crossover under the current VIVIT_TWO_STAGE setting."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels

def med(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3

for n in [int(a) for a in sys.argv[1:]] or [2304, 2560, 3072, 3584, 4096]:
    V = torch.randn(n, n // 2, device="cuda")
    G = V @ V.T
    tv = med(lambda: kernels.symeig(G.clone(), eigenvectors=False, overwrite=True))
    tz = med(lambda: kernels.symeig(G.clone(), eigenvectors=True, overwrite=True))
    print(f"n={n} two_stage={os.environ.get('VIVIT_TWO_STAGE', 'auto')}: eigvalsh {tv:.1f} ms, symeig {tz:.1f} ms", flush=True)
