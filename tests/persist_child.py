"""Child process of tests/test_persistent_gpu.py: the persistent kernels (sb2st bulge chasing, one-XCD sytrd, one-XCD panel
QR of sy2sb) under ONE setting of their knobs (VIVIT_SB2ST_PERSIST / VIVIT_SYTRD_PERSIST / VIVIT_QR_PERSIST are read once
per process); SHA-1 of the outputs that must be bit-identical, the others in full, as JSON.

usage: python persist_child.py OUT.json
"""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vivit_amd import kernels  # noqa: E402

NB = 64
DEV = torch.device("cuda:0")


def sha(*tensors):
    h = hashlib.sha1()
    for t in tensors:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def band(n, seed):
    g = torch.Generator().manual_seed(seed)
    AB = torch.randn(n, 2 * NB + 1, generator=g)
    AB[:, :NB] = 0
    for i in range(min(n, NB)):
        AB[i, : 2 * NB - i] = 0      # columns left of the matrix
    return AB


def main():
    out = {"sb2st": {}, "sytrd": {}}
    for n in (960, 1000, 1025, 2048, 4100):
        d, e, R2, tau2 = kernels.sb2st(band(n, n).to(DEV))
        nk = tau2.shape[1]
        # (the last column and the last two rows of tau2 are scratch of the persistent kernel)
        out["sb2st"][str(n)] = {"de": sha(d, e), "tau2": sha(tau2[: n - 2, : nk - 1]), "R2": sha(R2)}
    for n in (193, 256, 300, 777, 1024, 1280, 1500, 2048):
        g = torch.Generator().manual_seed(n)
        M = torch.randn(n, n, generator=g)
        S = (M + M.T).to(DEV)
        d, e, tau, A = kernels.sytrd(S)
        out["sytrd"][str(n)] = {"d": d.cpu().tolist(), "e": e.cpu().tolist()}
    out["sy2sb"] = {}
    for n in (200, 1000, 2500):
        g = torch.Generator().manual_seed(n)
        M = torch.randn(n, n, generator=g)
        S = ((M + M.T) / 2).to(DEV)
        AB, tau1, A = kernels.sy2sb(S)
        torch.save({"AB": AB.cpu(), "tau1": tau1.cpu()}, sys.argv[1] + f".sy2sb{n}.pt")
        out["sy2sb"][str(n)] = sys.argv[1] + f".sy2sb{n}.pt"
    with open(sys.argv[1], "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
