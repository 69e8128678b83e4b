"""Time of the sliding-window Q2 kernel alone at one size (mode 1 only; for variant libraries and profilers).
usage: python scripts/probe/q2_time1.py n [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vivit_amd import kernels  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from q2_time import band, timed, DEV  # noqa: E402

n = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d, e, R2, tau2 = kernels.sb2st(band(n))
Z = torch.zeros(n, n, device=DEV)
Z.diagonal().fill_(1.0)
kernels.q2_apply_(Z, R2, tau2, mode=1)
t, ts = timed(lambda: kernels.q2_apply_(Z, R2, tau2, mode=1), reps)
print(f"n={n} lib={os.environ.get('VIVIT_HIP_LIB', 'product')} waves={os.environ.get('VIVIT_Q2_SLIDE_WAVES', 'auto')}: "
      f"{t * 1e3:.1f} ms {['%.1f' % (x * 1e3) for x in ts]}", flush=True)
