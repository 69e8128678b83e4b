#!/usr/bin/env python3
"""Headline benchmark: GGN eigenpairs/sec (Gram build + symeig), MLP 784-512-10, batch 4096, fp32.

One step = one pass of the hot path over one batch of synthetic input:
    G = sum_p V_p V_p^T  (4 SYRK launches on MFMA, materialised sqrt-GGN factors, n = N*C = 40960)
    (w, Z) = symeig(G)   (all n eigenpairs: tridiagonalisation + divide & conquer + back-transform)
The factors V_p (what BackPACK's SqrtGGNExact attaches to the parameters, 66.7 GB) are resident in
HBM before the timed region starts; they are synthetic (random-init MLP, uniform random inputs).

Multi-GPU (`torchrun --nproc-per-node N bench.py --gpus N ...`): the SAME global problem, the
contraction (parameter) dimension of V is sharded across ranks, every rank builds a partial Gram
matrix, the partial Grams are summed with one RCCL all-reduce over xGMI; the eigensolver's reduction
and tridiagonal solve run replicated (deterministic: all ranks hold identical intermediates), its
back-transformations (independent per eigenvector) are sharded and the eigenvector slices all-gathered.
That is strong scaling; `phases` reports the Gram / all-reduce / symeig split so the Gram-build
scaling can be read off directly.

Prints ONE JSON line on rank 0 (contract: see DESIGN.md section "Measurement").
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
# HBM-side traffic of ONE launch of the dominant kernel (the first layer's Gram SYRK, n = 40960, P = 401408) from
# separate `rocprofv3 --pmc` passes (profiles/r01_pmc/syrk256_n40960_p401408_*.csv, corrected as the guide
# prescribes: 2 x FETCH_SIZE KiB + WRITE_SIZE KiB; FETCH_SIZE includes Infinity-Cache hits, so this is an upper
# bound).  PMC collection cannot run inside this process; the constant is only attached to the exact shape it
# was measured on.
SYRK_TRAFFIC_BYTES_PMC = {("mlp784-512-10_b4096", 1): (2 * 3695931059.375 + 276639882.25) * 1024.0}
MFMA_F32_PEAK_TF = 157.3  # dense fp32 MFMA peak (same guide)

WORKLOADS = {
    # name: (layers, batch, classes)
    "mlp784-512-10_b4096": ((784, 512, 10), 4096, 10),
    "mlp784-512-10_b128": ((784, 512, 10), 128, 10),   # BASELINE config 1 (CPU-runnable)
    "mlp784-512-10_b1024": ((784, 512, 10), 1024, 10),
}


def mlp_sqrt_ggn_factors(dims, batch, device, shard=(0, 1), seed=0):
    """Materialised exact sqrt-GGN factors of Sequential(Linear, ReLU, Linear) + CrossEntropy(mean).

    Returns a list of [n, P_local] matrices (n = C*batch, class-major rows), one per parameter
    (slice): exactly what BackPACK's SqrtGGNExact stores in ``param.sqrt_ggn_exact`` viewed 2-D.
    ``shard=(r, R)``: rank r keeps the r-th of R slices of the first layer's output units for the
    big first-layer weight; the small parameters go to rank 0.  Synthetic-input generator, outside
    the timed region (torch ops).
    """
    d_in, d_h, C = dims
    r, R = shard
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)  # default PyTorch init ("random weight"), X = rand as in test/settings.py:31
        lin1 = torch.nn.Linear(d_in, d_h)
        lin2 = torch.nn.Linear(d_h, C)
        X = torch.rand(batch, d_in)
    W1, b1, W2, b2 = (t.detach().to(device) for t in (lin1.weight, lin1.bias, lin2.weight, lin2.bias))
    X = X.to(device)
    with torch.no_grad():
        z1 = X @ W1.T + b1
        a1 = z1.clamp_min(0)
        out = a1 @ W2.T + b2
        p = out.softmax(1)
        sq = p.sqrt()
        eye = torch.eye(C, device=device)
        S = (torch.einsum("nv,vc->vnc", sq, eye) - torch.einsum("nv,nc->vnc", sq, p)) / math.sqrt(batch)  # [C,N,C]
        n = C * batch
        facs = []
        # layer 2 (rank 0 only)
        if r == 0:
            facs.append(torch.einsum("vno,ni->vnoi", S, a1).reshape(n, -1))  # W2: [n, C*d_h]
            facs.append(S.reshape(n, C).clone())                            # b2
        M1 = (S @ W2) * (z1 > 0).unsqueeze(0)                               # [C, N, d_h]
        lo, hi = (d_h * r) // R, (d_h * (r + 1)) // R
        facs.append(torch.einsum("vno,ni->vnoi", M1[:, :, lo:hi], X).reshape(n, -1))  # W1 slice
        if r == 0:
            facs.append(M1.reshape(n, d_h).clone())                         # b1
    return facs


def cpu_baseline(dims, C, sample_batch, full_n, full_P):
    """The oracle (CPU restatement of the reference algorithm: einsum Gram over every parameter with
    the full 2 n^2 P work, vivit/utils/gram.py:230-232,104-116, then torch.linalg.eigh, successor of
    the Tensor.symeig call at vivit/linalg/eigh.py:248-250) timed on this host on a bounded sample."""
    from oracle import vivit_oracle as oracle

    cores = torch.get_num_threads()
    facs = mlp_sqrt_ggn_factors(dims, sample_batch, torch.device("cpu"))
    n = facs[0].shape[0]
    V = [f.view(C, sample_batch, -1) for f in facs]
    t0 = time.perf_counter()
    gram = oracle.compute_gram_mat(V, start_dim=2, flatten=True)
    t_gram = time.perf_counter() - t0
    t0 = time.perf_counter()
    evals, evecs = oracle.tensor_symeig(gram, eigenvectors=True)
    t_eig = time.perf_counter() - t0
    sample_rate = n / (t_gram + t_eig)
    # scale to the full workload with the textbook cost model (Gram ~ n^2 P, eigh ~ n^3)
    est = full_n / (t_gram * (full_n / n) ** 2 + t_eig * (full_n / n) ** 3)
    return {
        "value": est,
        "unit": "eigenpairs/s",
        "cores": cores,
        "kind": "port",
        "sample": (
            f"same MLP at batch {sample_batch} (n={n}, P={full_P}): einsum Gram {t_gram:.2f} s + torch.linalg.eigh "
            f"{t_eig:.2f} s = {sample_rate:.1f} eigenpairs/s measured; value = that sample scaled to n={full_n} with "
            f"Gram ~ n^2, eigh ~ n^3"
        ),
        "sample_value": sample_rate,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mlp784-512-10_b4096", choices=sorted(WORKLOADS))
    ap.add_argument("--values-only", action="store_true", help="EigvalshComputation flavour (no eigenvectors)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=256)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP device); there is no CPU fallback")
    # (functional testing on a 1-GPU box: VIVIT_DIST_BACKEND=gloo lets several ranks share device 0)
    backend = os.environ.get("VIVIT_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from vivit_amd import _lib, kernels
    from vivit_amd import distributed as vdist

    dims, batch, C = WORKLOADS[args.workload]
    n = C * batch
    P_total = dims[0] * dims[1] + dims[1] + dims[1] * dims[2] + dims[2]
    facs = mlp_sqrt_ggn_factors(dims, batch, device, shard=(rank, world))
    p_local = sum(f.shape[1] for f in facs)
    G = torch.empty((n, n), dtype=torch.float32, device=device)
    lib = _lib.load()
    vectors = not args.values_only

    ev = {k: [torch.cuda.Event(enable_timing=True) for _ in range(4)] for k in range(args.steps + args.warmup)}

    def step(idx):
        e = ev[idx]
        e[0].record()
        for k, A in enumerate(facs):
            kernels.gram_syrk(A, out=G, alpha=1.0, beta=0.0 if k == 0 else 1.0)
        e[1].record()
        if dist is not None:
            dist.all_reduce(G)
        e[2].record()
        if dist is not None and vectors:
            # reduction + tridiagonal solve replicated, back-transformations sharded by eigenvector, all-gather
            w, Z = vdist.symeig(G, overwrite=True)
        else:
            w, Z = kernels.symeig(G, eigenvectors=vectors, overwrite=True)
        e[3].record()
        return w, Z

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    import ctypes

    lib.vivit_profile_begin(64)
    t0 = time.perf_counter()
    for i in range(args.steps):
        w, Z = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = (ctypes.c_double * 6)()
    lib.vivit_profile_end(prof)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gram_s = sum(ev[args.warmup + i][0].elapsed_time(ev[args.warmup + i][1]) for i in range(args.steps)) / 1e3 / args.steps
    ar_s = sum(ev[args.warmup + i][1].elapsed_time(ev[args.warmup + i][2]) for i in range(args.steps)) / 1e3 / args.steps
    eig_s = sum(ev[args.warmup + i][2].elapsed_time(ev[args.warmup + i][3]) for i in range(args.steps)) / 1e3 / args.steps

    if rank == 0:
        value = n * args.steps / elapsed
        syrk_cnt, syrk_ms, syrk_flops, symv_cnt, symv_ms, symv_bytes = list(prof)
        symv_total_s = (symv_ms / max(symv_cnt, 1)) * (n - 2) / 1e3  # sampled mean x launches per step
        syrk_total_s = syrk_ms / 1e3 / max(args.steps, 1)
        if symv_cnt > 0 and symv_total_s >= syrk_total_s:
            achieved = symv_bytes / (symv_ms / 1e3) / 1e9
            roofline = {
                "kernel": "trd_symv_kernel (tridiagonalisation, lower-triangle symmetric matrix-vector product)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "launches_sampled": int(symv_cnt), "avg_launch_ms": symv_ms / max(symv_cnt, 1),
                "est_share_of_step": symv_total_s / (elapsed / args.steps),
            }
        else:
            achieved = syrk_flops / (syrk_ms / 1e3) / 1e12
            roofline = {
                "kernel": "gemm256_kernel<LAY_K,LAY_K> (Gram SYRK, fp32 MFMA)",
                "bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F32_PEAK_TF,
                "traffic": SYRK_TRAFFIC_BYTES_PMC.get((args.workload, world)),
                "traffic_note": "bytes of the first-layer weight's SYRK launch (98.6 % of the Gram flops, 4.73 s), separate "
                                "rocprofv3 --pmc passes (profiles/r01_pmc), FETCH_SIZE includes Infinity-Cache hits",
                "launches_sampled": int(syrk_cnt), "avg_launch_ms": syrk_ms / max(syrk_cnt, 1),
                "est_share_of_step": syrk_total_s / (elapsed / args.steps),
            }
        secondary = {
            "gram_syrk_tflops": (syrk_flops / (syrk_ms / 1e3) / 1e12) if syrk_ms > 0 else None,
            "symv_gbs": (symv_bytes / (symv_ms / 1e3) / 1e9) if symv_ms > 0 else None,
        }
        out = {
            "metric": "GGN eigenpairs/sec (Gram build + symeig), MLP 784-512-10, batch=4096",
            "value": value,
            "unit": "eigenpairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded random-init MLP, uniform random inputs, materialised exact sqrt-GGN factors)",
            "config": {
                "workload": args.workload,
                "n": n, "P": P_total, "P_local_rank0": p_local, "eigenvectors": vectors,
                "pairs_per_step": n,
                "parallelism": (f"parameter-sharded Gram x{world} + RCCL all-reduce; symeig: replicated reduction/D&C, "
                                f"back-transformations sharded x{world} + all-gather") if world > 1 else "single GPU",
            },
            "phases": {"gram_s": gram_s, "allreduce_s": ar_s, "symeig_s": eig_s},
            "roofline": roofline,
            "kernels": secondary,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dims, C, min(args.cpu_sample_batch, batch), n, P_total)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
