"""Child process of tests/test_gram_precision_gpu.py: the Gram SYRK under ONE setting of VIVIT_GEMM_SPLIT (the library
reads the knob once per process), error statistics against fp64 dot products written as JSON.

Errors are reported in RANDOM-WALK units: for rows with independent zero-mean entries the exact product a_i . a_j is a
sum of K terms of size rms_i rms_j, so ``|err| / (rms_i rms_j sqrt(K))`` is the error of the kernel's arithmetic per
term, independent of K and of the row scales.  A kernel that drops the 2^-16 partial products of the three-way bf16
split shows ~1e-5 here at every K; fp32 accumulation rounding is a few 1e-7.

usage: python gram_precision_child.py OUT.json CASE [CASE ...]     CASE = n:K[:mixed]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vivit_amd import _lib, kernels  # noqa: E402


def rows_for(n, per_class=24, seed=0):
    """Sample rows covering every 256-row tile class: first/last tile, tile boundaries, random interior."""
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, n, (per_class,), generator=g).tolist()
    idx += [0, 1, 255, 256, 257, n - 1, n - 2, n // 2, n // 2 + 1]
    return sorted(set(i for i in idx if 0 <= i < n))


def run_case(n, K, mixed, dev):
    g = torch.Generator(device=dev).manual_seed(1234 + n + K)
    A = torch.randn((n, K), generator=g, device=dev, dtype=torch.float32)
    scale = torch.ones(n, device=dev, dtype=torch.float64)
    if mixed:  # rows of magnitude 2^-40, 1, 2^+40 (exact power-of-two scalings)
        e = (torch.arange(n, device=dev) % 3 - 1).double() * 40.0
        scale = torch.pow(torch.tensor(2.0, dtype=torch.float64, device=dev), e)
        A.mul_(scale.float().unsqueeze(1))
    G = kernels.gram_syrk(A)
    torch.cuda.synchronize()
    ms = None
    if os.environ.get("VIVIT_PREC_TIME"):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(3):
            kernels.gram_syrk(A, out=G)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 3
    ri = torch.tensor(rows_for(n, seed=1), device=dev)
    ci = torch.tensor(rows_for(n, seed=2), device=dev)
    ref = A[ri].double() @ A[ci].double().T                                  # [R, C] fp64
    got = G[ri][:, ci].double()
    got_t = G[ci][:, ri].double().T                                          # mirrored entries
    rms = A.double().pow(2).mean(1).sqrt()
    unit = rms[ri].unsqueeze(1) * rms[ci].unsqueeze(0) * (K ** 0.5)
    off = ri.unsqueeze(1) != ci.unsqueeze(0)
    serr = ((got - ref) / unit)[off]
    sref = (ref / unit)[off]
    slope = float((serr * sref).sum() / (sref * sref).sum())   # err ~ slope * ref: a relative shrink/growth of every entry
    err = ((got - ref).abs() / unit)[off]
    err_t = ((got_t - ref).abs() / unit)[off]
    # diagonal: a sum of squares (all terms positive), error relative to the entry itself
    d = torch.arange(n, device=dev)
    dref = A.double().pow(2).sum(1)
    derr = ((G[d, d].double() - dref).abs() / dref)
    dmean = float(((G[d, d].double() - dref) / dref).mean())
    return {
        "n": n, "K": K, "mixed": bool(mixed),
        "offdiag_rms": float(err.pow(2).mean().sqrt()), "offdiag_max": float(err.max()),
        "mirror_rms": float(err_t.pow(2).mean().sqrt()), "mirror_max": float(err_t.max()),
        "diag_rms": float(derr.pow(2).mean().sqrt()), "diag_max": float(derr.max()), "diag_mean": dmean,
        "offdiag_slope": slope, "ms": ms,
        "symmetric": bool(torch.equal(G, G.T)),
        "finite": bool(torch.isfinite(G).all()),
        "entries": int(off.sum()),
    }


def main():
    out, cases = sys.argv[1], sys.argv[2:]
    dev = torch.device("cuda:0")
    res = {"split_mode": int(_lib.load().vivit_gemm_split_mode()), "cases": []}
    for c in cases:
        parts = c.split(":")
        n, K = int(parts[0]), int(parts[1])
        res["cases"].append(run_case(n, K, len(parts) > 2 and parts[2] == "mixed", dev))
        torch.cuda.empty_cache()
    with open(out, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
