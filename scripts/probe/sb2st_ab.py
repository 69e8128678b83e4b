"""sb2st: persistent kernel vs launch chain, bitwise (same arithmetic -> d, e, tau2 must be identical).  Run once per
setting (the knobs are read once per process): prints a checksum line per n."""
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from vivit_amd import kernels
NB = 64
for n in [int(a) for a in sys.argv[1:]] or [1000, 2048, 4100]:
    g = torch.Generator().manual_seed(n)
    AB = torch.randn(n, 2 * NB + 1, generator=g)
    AB[:, :NB] = 0
    for i in range(min(n, NB)):
        AB[i, : 2 * NB - i] = 0
    d, e, R2, tau2 = kernels.sb2st(AB.to("cuda"))
    torch.cuda.synchronize()
    h = hashlib.sha1(d.cpu().numpy().tobytes() + e.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"n={n} persist={os.environ.get('VIVIT_SB2ST_PERSIST', '1')} wt={os.environ.get('VIVIT_SB2ST_WT', '0')} d/e sha {h} |d|max {float(d.abs().max()):.4f}", flush=True)
