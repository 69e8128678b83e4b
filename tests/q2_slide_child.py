"""Child process of tests/test_q2_slide_gpu.py::test_slide_wave_configurations: the sliding-window Q2 kernel under ONE setting
of VIVIT_Q2_SLIDE_WAVES / VIVIT_Q2_SLIDE_LOADERS (read once per process).  With LOADERS=0 -- or more than ten compute waves --
the compute waves request the block images themselves (the "self-load" path with its hand-counted s_waitcnt vmcnt(4) waits,
csrc/q2slide.hip); the default run of the test suite never gets there (it needs > 160 rows per CU).  Errors against the
fp64 sequential reflectors (small n) and the block-step kernels (mode 0), as JSON.

usage: python q2_slide_child.py OUT.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_q2_slide_gpu import DEV, apply_reference, chase  # noqa: E402
from vivit_amd import kernels  # noqa: E402


def main():
    nw = int(os.environ.get("VIVIT_Q2_SLIDE_WAVES", "0"))
    out = {"sequential": {}, "block_steps": {}}
    # (n % 64 != 0 throughout; row counts that are not a multiple of the 16 nw rows of a slab: ragged last slab and wave)
    for n, nrows in ((324, 16 * nw + 5), (516, 2 * 16 * nw + 37), (452, 3 * 16 * nw)):
        band, d, e, R2, tau2 = chase(n, n)
        rng = np.random.default_rng(n + nrows)
        Z0 = (rng.standard_normal((nrows, n)) / np.sqrt(n)).astype(np.float32)
        ref = apply_reference(Z0, R2.cpu().numpy(), tau2.cpu().numpy())
        got = kernels.q2_apply_(torch.from_numpy(Z0).to(DEV).clone(), R2, tau2, mode=1).cpu().double().numpy()
        old = kernels.q2_apply_(torch.from_numpy(Z0).to(DEV).clone(), R2, tau2, mode=0).cpu().double().numpy()
        scale = np.abs(ref).max()
        out["sequential"][f"{n}x{nrows}"] = {"slide": float(np.abs(got - ref).max() / scale), "block": float(np.abs(old - ref).max() / scale)}
    for n, nrows in ((2052, 5 * 16 * nw + 123), (1220, 40 * 16 * nw + 16)):
        band, d, e, R2, tau2 = chase(n, n)
        g = torch.Generator().manual_seed(n)
        Q = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))[0].float()
        reps = -(-nrows // n)
        Z0 = torch.cat([Q * (1.0 if r % 2 == 0 else -1.0) for r in range(reps)], 0)[:nrows].contiguous().to(DEV)
        new = kernels.q2_apply_(Z0.clone(), R2, tau2, mode=1)
        new2 = kernels.q2_apply_(Z0.clone(), R2, tau2, mode=1)
        old = kernels.q2_apply_(Z0.clone(), R2, tau2, mode=0)
        rows = new[: min(nrows, n)].double()
        orth = float((rows @ rows.T - torch.eye(rows.shape[0], dtype=torch.float64, device=rows.device)).abs().max())
        out["block_steps"][f"{n}x{nrows}"] = {"diff": float((new - old).abs().max() / old.abs().max()), "orth": orth,
                                              "bitwise_repeat": bool(torch.equal(new, new2))}
    with open(sys.argv[1], "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
