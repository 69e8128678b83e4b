"""Q2 back-transformation alone: block-step kernels (mode 0) against the sliding-window kernel (mode 1) on the
reflectors of a random band matrix; time (HIP events, median of 3) and orthonormality of Q2^T itself (Zt = I).
usage: python scripts/probe/q2_time.py [n ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vivit_amd import kernels  # noqa: E402

DEV = torch.device("cuda:0")
NB = 64


def band(n, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    AB = torch.randn(n, 2 * NB + 1, generator=g, device=DEV)
    AB[:, :NB] = 0
    for i in range(min(n, NB)):
        AB[i, : 2 * NB - i] = 0
    return AB


def timed(fn, reps=3):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 1e3)
    return sorted(ts)[len(ts) // 2], ts


def main(sizes):
  for n in sizes:
      d, e, R2, tau2 = kernels.sb2st(band(n))
      del d, e
      Z = torch.empty(n, n, device=DEV)
      res = {}
      for mode in (0, 1):
          def run():
              Z.zero_()
              Z.diagonal().fill_(1.0)
              kernels.q2_apply_(Z, R2, tau2, mode=mode)
          run()  # warm-up
          t, ts = timed(run)
          # orthonormality of the rows of Q2^T: fp64 accumulation on 256 sampled rows against all
          idx = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:256].to(DEV)
          acc = torch.zeros(256, n, dtype=torch.float64, device=DEV)
          Zs = Z[idx].double()
          for lo in range(0, n, 4096):
              acc += Zs[:, lo:lo + 4096] @ Z[:, lo:lo + 4096].double().T
          acc[torch.arange(256, device=DEV), idx] -= 1.0
          res[mode] = (t, acc.abs().max().item(), acc.pow(2).mean().sqrt().item(), Z.clone() if n <= 16384 else None)
          print(f"n={n} mode={mode}: {t * 1e3:.1f} ms {['%.1f' % (x * 1e3) for x in ts]}  orth max {res[mode][1]:.2e} rms {res[mode][2]:.2e}", flush=True)
      if res[0][3] is not None:
          print(f"n={n} max |mode1 - mode0| = {(res[1][3] - res[0][3]).abs().max().item():.2e}")
      del Z, R2, tau2, res
      kernels._WORKSPACES.clear()
      torch.cuda.empty_cache()


if __name__ == "__main__":
    main([int(x) for x in sys.argv[1:]] or [8192, 16384, 40960])
