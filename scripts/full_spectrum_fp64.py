#!/usr/bin/env python3
"""One-off: the FULL spectrum of the headline Gram matrix (BASELINE config 2: n = 40 960, P = 407 050) from the HIP path
against an fp64 checker -- every eigenvalue, not a property.

  * G32 = the product path's Gram matrix (4 SYRKs on the bf16 pipe, fp32), w32 = kernels.symeig(G32) (all eigenvectors) and
    kernels.symeig(G32, eigenvectors=False) (Sturm multisection);
  * checker A: torch.linalg.eigvalsh(G32.double()) on the device (rocSOLVER, fp64) -- isolates the EIGENSOLVER's error;
  * checker B: G64 = the same factors contracted in fp64 (slab by slab), eigvalsh(G64) -- the error of the whole path
    (Gram arithmetic + eigensolver) against "exact".
Reported per comparison: max |w - ref| / lambda_max over all n, over the top min(n, P) (the scope of
test/linalg/test_eigvalsh.py:55-60 of the reference), and the reference's own criterion
allclose(rtol=1e-4, atol=5e-6) on that range.  Writes ONE JSON line (profiles/r05_full_spectrum_fp64.json).

usage: python scripts/full_spectrum_fp64.py [--workload mlp784-512-10_b4096] [--out FILE] [--skip-b]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from vivit_amd import kernels  # noqa: E402


def compare(w, ref, k):
    w, ref = w.double(), ref.double()
    lam = ref[-1].item()
    d = (w - ref).abs()
    top = slice(w.numel() - k, w.numel())
    ok = bool(torch.isclose(w[top], ref[top], rtol=1e-4, atol=5e-6).all())
    return {"max_abs_err_over_lambda_max_all": d.max().item() / lam, "max_abs_err_over_lambda_max_top": d[top].max().item() / lam,
            "rms_abs_err_over_lambda_max": d.pow(2).mean().sqrt().item() / lam,
            "reference_allclose_rtol1e-4_atol5e-6_top": ok, "argmax": int(d.argmax()), "lambda_max_ref": lam}


def eigvalsh_fp64(G64, tag):
    """fp64 eigenvalues of a symmetric fp64 device matrix: rocSOLVER through torch on the device where it works (it refuses
    n = 40 960: hipsolverDnDsyevd_bufferSize -> HIPSOLVER_STATUS_INTERNAL_ERROR), host LAPACK (torch CPU, 16 threads) otherwise."""
    try:
        w = torch.linalg.eigvalsh(G64)
        torch.cuda.synchronize()
        return w, "device (rocSOLVER dsyevd)"
    except RuntimeError as exc:
        print(f"[full-spectrum] {tag}: device eigvalsh failed ({str(exc).splitlines()[0][:120]}); host LAPACK ...", file=sys.stderr, flush=True)
    torch.set_num_threads(int(os.environ.get("VIVIT_CHECK_THREADS", "16")))
    host = G64.cpu()
    w = torch.linalg.eigvalsh(host)
    del host
    return w.to(G64.device), f"host LAPACK dsyevd ({torch.get_num_threads()} threads)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mlp784-512-10_b4096")
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip-b", action="store_true", help="skip the fp64 Gram matrix (checker B)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dims, batch, C = bench.WORKLOADS[args.workload]
    n = C * batch
    P = dims[0] * dims[1] + dims[1] + dims[1] * dims[2] + dims[2]
    k = min(n, P)
    facs = bench.mlp_sqrt_ggn_factors(dims, batch, dev)
    G = torch.empty((n, n), dtype=torch.float32, device=dev)
    for i, A in enumerate(facs):
        kernels.gram_syrk(A, out=G, alpha=1.0, beta=0.0 if i == 0 else 1.0)
    out = {"workload": args.workload, "n": n, "P": P, "top": k}
    G64b = None
    if not args.skip_b:
        # checker B's matrix first (needs the factors): fp64 contraction in row slabs, lower block triangle + mirror
        t0 = time.perf_counter()
        G64b = torch.zeros((n, n), dtype=torch.float64, device=dev)
        B = 2048
        for A in facs:
            cols = 32768
            for c0 in range(0, A.shape[1], cols):
                Ad = A[:, c0:c0 + cols].double()
                for i in range(0, n, B):
                    G64b[i:i + B, : i + B] += Ad[i:i + B] @ Ad[: i + B].T
                del Ad
        iu = torch.triu_indices(n, n, 1, device=dev) if n <= 8192 else None
        if iu is not None:
            G64b[iu[0], iu[1]] = G64b[iu[1], iu[0]]
        else:
            for i in range(0, n, B):   # mirror block rows
                G64b[i:i + B, i + B:] = G64b[i + B:, i:i + B].T
                blk = G64b[i:i + B, i:i + B]
                G64b[i:i + B, i:i + B] = torch.tril(blk) + torch.tril(blk, -1).T
        torch.cuda.synchronize()
        out["gram_fp64_s"] = time.perf_counter() - t0
        out["gram_entry_err_vs_fp64_over_lambda_scale"] = ((G.double() - G64b).abs().max() / G64b.diagonal().max()).item() if n <= 16384 else None
    del facs
    torch.cuda.empty_cache()
    print(f"[full-spectrum] Gram matrices ready (n = {n})", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    w_dc, _ = kernels.symeig(G, eigenvectors=True)
    w_st, _ = kernels.symeig(G, eigenvectors=False)
    torch.cuda.synchronize()
    out["hip_symeig_both_s"] = time.perf_counter() - t0
    torch.cuda.empty_cache()
    print("[full-spectrum] HIP solves done; fp64 eigvalsh of the fp32 Gram matrix (checker A) ...", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    refA, how = eigvalsh_fp64(G.double(), "checker A")
    out["checker_a_s"] = time.perf_counter() - t0
    out["checker_a"] = how
    out["eigensolver_vs_fp64_eigvalsh_of_same_matrix"] = {"symeig_all_vectors": compare(w_dc, refA, k), "eigvalsh_multisection": compare(w_st, refA, k)}
    print(f"[full-spectrum] checker A done in {out['checker_a_s']:.1f} s: {out['eigensolver_vs_fp64_eigvalsh_of_same_matrix']['symeig_all_vectors']}",
          file=sys.stderr, flush=True)
    if G64b is not None:
        del G
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        refB, how = eigvalsh_fp64(G64b, "checker B")
        out["checker_b_s"] = time.perf_counter() - t0
        out["checker_b"] = how
        out["whole_path_vs_fp64_gram_fp64_eigvalsh"] = {"symeig_all_vectors": compare(w_dc, refB, k), "eigvalsh_multisection": compare(w_st, refB, k)}
    line = json.dumps(out)
    print(line, flush=True)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
