#!/usr/bin/env python3
"""Headline benchmark: GGN eigenpairs/sec (Gram build + symeig), MLP 784-512-10, batch 4096, fp32.

One step = one pass of the hot path over one batch of synthetic input:
    G = sum_p V_p V_p^T  (4 SYRK launches on MFMA, materialised sqrt-GGN factors, n = N*C = 40960)
    (w, Z) = symeig(G)   (all n eigenpairs: tridiagonalisation + divide & conquer + back-transform)
The factors V_p (what BackPACK's SqrtGGNExact attaches to the parameters, 66.7 GB) are resident in
HBM before the timed region starts; they are synthetic (random-init MLP, uniform random inputs).

Multi-GPU (`python bench.py --gpus N` spawns its N ranks itself; under torchrun it uses the ranks it is
given): the SAME global problem, DATA parallel -- rank g holds the factors of its batch shard only
(`[C, N/R, P]`, what a per-GPU backward pass leaves behind).  The Gram matrix couples all sample pairs, so
one all-to-all per parameter first turns batch shards into parameter shards (all 7 xGMI links of every GPU
at once, V moves once), every rank runs the full-size SYRK over 1/R of the contraction length, and the
partial Gram matrices are summed with one RCCL all-reduce (vivit_amd/distributed.py).  The eigensolver's
reduction and tridiagonal solve run replicated (deterministic: all ranks hold identical intermediates), its
back-transformations (independent per eigenvector) are sharded and the eigenvector slices all-gathered.
That is strong scaling; `phases` reports the exchange / Gram / all-reduce / symeig split so the Gram-build
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
# the same launch on the bf16-pipe path (profiles/r03_pmc/v3_syrkbx_n40960_p401408_*.csv, final code of round 3: 7 chunk
# launches of gemm256_bx_kernel<6> + 7 of bx_split_kernel, summed; the in-loop chain flushes add the Gram matrix's own
# read-modify-write traffic: 6.16 TB fetched / 0.61 TB written; with 512-k chains on the diagonal tiles it was 7.74 / 0.66,
# round 2 on 8192-k chains: 5.97 / 0.27)
# Round 4 (profiles/r04_pmc/pmc_syrk_*, final code: one 4096-column chain per launch, 98 launches of each kernel): 4.76 TB
# fetched / 0.83 TB written -- the workgroups of an XCD start every chain together and share their operand panels again.
# Round 5 (profiles/r05_pmc/pmc_syrk_*, final code: the same 98 + 98 launches): 4.78 TB fetched / 0.83 TB written.
# Round 6 (profiles/r06_pmc/pmc_syrk_*, final code: the asm K loop, the same 98 + 98 launches): FETCH_SIZE 2.3295e9 + 3.2237e7 KiB,
# WRITE_SIZE 7.9816e8 + 9.6339e7 KiB = 4.84 TB fetched (with the wide-read correction) / 0.92 TB written.
# Round 6, second half (profiles/r06_pmc_s16/pmc_syrk_*, final code: 16x16x32 MFMAs, register flush, the same 98 + 98 launches): FETCH_SIZE
# 2.4568e9 + 3.2238e7 KiB, WRITE_SIZE 6.3943e8 + 9.6339e7 KiB = 5.10 TB fetched (with the wide-read correction) / 0.75 TB written.
SYRK_BX_TRAFFIC_BYTES_PMC = {("mlp784-512-10_b4096", 1): (2 * (2.4568e9 + 3.2238e7) + (6.3943e8 + 9.6339e7)) * 1024.0}
# clock the chip holds under that kernel, measured OUTSIDE this process (the product library carries no stamps):
# in-kernel s_memtime / s_memrealtime stamps of a diagnostic build (scripts/probe/bx_clock.py + libstamp.so, median over
# the 12 880 workgroups of the last chunk launch after 6 s of back-to-back SYRKs on the bench's own factors;
# profiles/r03_pmc/v3_bx_clock_real.txt) and GRBM_GUI_ACTIVE / 8 / duration of the PMC pass on N(0,1) data
# (round 4, profiles/r04_pmc/pmc_syrk_mfma_*: GRBM_GUI_ACTIVE 4.1806e10 / 8 over 2.986 s of gemm256_bx_kernel on N(0,1) data,
# SQ_VALU_MFMA_BUSY_CYCLES 3.9707e12 over 1024 SIMDs; the in-kernel stamps are round 3's)
# round 5 (profiles/r05_pmc/pmc_syrk_mfma_*: the same 98 launches on N(0,1) data, final code = global -> LDS requests issued in row 3):
# GRBM_GUI_ACTIVE 4.0016e10 / 8 over 2.807 s = 1.782 GHz, SQ_VALU_MFMA_BUSY_CYCLES 3.9707e12 over 1024 SIMDs = 77.5 % of the cycles
# (before the request placement, same round: 4.2170e10 / 8 over 2.951 s = 1.786 GHz, 73.6 %)
# and on the benchmark's OWN first-layer factor (scripts/pmc_syrk_full.py bench; profiles/r05_pmc/pmc_syrk_mfma_bench_*): GRBM_GUI_ACTIVE 4.0224e10 / 8
# over 2.520 s = 1.996 GHz, SQ_VALU_MFMA_BUSY_CYCLES 3.9707e12 = 77.1 % of the cycles: 267.3 TFLOP/s = 0.771 x 1.996 / 2.4 = 0.641 of the ceiling
# round 6 (profiles/r06_pmc/pmc_syrk_mfma_*, profiles/r06_pmc_summary.txt; final code = the hand-scheduled K loop): N(0,1) data: GRBM_GUI_ACTIVE
# 3.7763e10 / 8 over 2.771 s = 1.703 GHz, SQ_VALU_MFMA_BUSY_CYCLES 3.9707e12 over 1024 SIMDs = 82.1 % of the cycles; the bench's own factor:
# 3.7714e10 / 8 over 2.403 s = 1.962 GHz, 82.3 %: 280.3 TFLOP/s = 0.823 x 1.962 / 2.4 = 0.672 of the ceiling (round 5: 77.1 % at 1.996 GHz = 0.641).
# The pipe is busier (77 -> 82 %) and the chip answers with a lower clock (DVFS give-back): - 7 % cycles of the K loop are - 3.8 % of its time.
# round 6, second half (profiles/r06_pmc_s16/pmc_syrk_mfma_*, profiles/r06_pmc_s16_summary.txt; final code = v_mfma_f32_16x16x32_bf16 with two partial
# products fused per instruction + the register flush): N(0,1) data: GRBM_GUI_ACTIVE 3.7401e10 / 8 over 2.621 s = 1.784 GHz, 82.9 % of the cycles;
# the bench's own factor: 3.7744e10 / 8 over 2.268 s = 2.080 GHz, 82.2 %: 297.0 TFLOP/s = 0.822 x 2.080 / 2.4 = 0.712 of the ceiling.  Same busy
# fraction as the 32x32x16 form above (the K loop needs 5 % MORE cycles, the flush fewer), but the chip holds a 5-6 % higher clock on this shape.
SYRK_BX_CLOCK_GHZ = {("mlp784-512-10_b4096", 1): {"in_kernel_stamps_half_zero_k_loop": 2.1, "in_kernel_stamps_randn_k_loop": 1.8,
                                                  "pmc_grbm_gui_active_randn": 1.784, "pmc_grbm_gui_active_bench_factors": 2.080, "nominal": 2.4,
                                                  "mfma_pipe_busy_pmc": 0.829, "mfma_pipe_busy_pmc_bench_factors": 0.822}}
# What the SAME per-K-tile instruction mix reaches with every byte of data movement removed (operand pieces in LDS once; no
# DMA, barrier, flush): scripts/probe/bx_bare_loop.hip, profiles/r05_bx_bare_loop.log -- fraction of the bf16 / 6 ceiling and the
# clock the chip holds, by operand data.  The power limit, not the kernel, takes the rest of the nominal peak.
# That probe is the 32x32x16 instruction mix.  For the product's 16x16x32 mix the same question is answered by the product's own asm block
# built without requests and without the barrier (timing-only variant U3 of scripts/gen_bx_kloop.py, in-kernel stamps,
# profiles/r06_bx16_timeline.log): 3351 core cycles per K tile at 2.4 GHz on half-zero data = 2 * 256^2 * 16 flop / 1.396 us x 256 CUs.
BX_BARE_LOOP_CEILING = {"mfma_32x32x16_mix": {"randn": {"frac": 0.708, "tflops_fp32_equiv": 296.8, "clock_ghz": 1.778},
                                              "half_zeros_like_the_bench_factors": {"frac": 0.787, "tflops_fp32_equiv": 330.0, "clock_ghz": 1.969},
                                              "zeros": {"frac": 0.976, "tflops_fp32_equiv": 409.3, "clock_ghz": 2.383}},
                        "half_zeros_like_the_bench_factors": {"frac": 0.917, "tflops_fp32_equiv": 384.6, "clock_ghz": 2.4, "cycles_per_k_tile": 3351},
                        "source": "the product's 16x16x32 K loop without requests and barrier (variant U3, profiles/r06_bx16_timeline.log); 32x32x16 mix: "
                                  "scripts/probe/bx_bare_loop.hip on one MI355X, 3 s per data kind, all 256 CUs (profiles/r05_bx_bare_loop.log, "
                                  "profiles/r06_bx_bare_loop_s16_v1.log)"}
MFMA_F32_PEAK_TF = 157.3  # dense fp32 MFMA peak (same guide)
MFMA_BF16_PEAK_TF = 2516.6  # dense bf16 MFMA peak (256 CU x 4 SIMD x 1024 flop/cycle x 2.4 GHz; same guide)

WORKLOADS = {
    # name: (layers, batch, classes)
    "mlp784-512-10_b4096": ((784, 512, 10), 4096, 10),
    "mlp784-512-10_b128": ((784, 512, 10), 128, 10),   # BASELINE config 1 (CPU-runnable)
    "mlp784-512-10_b1024": ((784, 512, 10), 1024, 10),
}


# the committed rocprofv3 outputs behind roofline.traffic / clock_ghz / clock_note (separate --pmc passes + kernel trace of the same command)
PMC_FILES = {
    "traffic": ["profiles/r06_pmc_s16/pmc_syrk_FETCH_SIZE_counter_collection.csv", "profiles/r06_pmc_s16/pmc_syrk_WRITE_SIZE_counter_collection.csv",
                "profiles/r06_pmc_s16/pmc_syrk_FETCH_SIZE_kernel_trace.csv", "profiles/r06_pmc_s16/pmc_syrk_WRITE_SIZE_kernel_trace.csv"],
    "clock_and_pipe_busy_randn": ["profiles/r06_pmc_s16/pmc_syrk_mfma_counter_collection.csv", "profiles/r06_pmc_s16/pmc_syrk_mfma_kernel_trace.csv"],
    "clock_and_pipe_busy_bench_factor": ["profiles/r06_pmc_s16/pmc_syrk_mfma_bench_counter_collection.csv",
                                         "profiles/r06_pmc_s16/pmc_syrk_mfma_bench_kernel_trace.csv"],
    "summary": "profiles/r06_pmc_s16_summary.txt (scripts/r06_pmc_summary.py over the passes above)",
    "command": "scripts/r06_measure.sh (rocprofv3 --pmc <counter> --kernel-trace -- python3 scripts/pmc_syrk_full.py [bench])",
    "kernel_trace_of_the_bench": "profiles/r06_bench_n40960_s16_kernel_stats.csv",
    "in_kernel_cycles_per_k_tile": "profiles/r06_bx16_timeline.log (scripts/probe/bx_timeline.py on -DBX_STAMP=2 builds)",
}


def mlp_sqrt_ggn_factors(dims, batch, device, shard=(0, 1), seed=0, samples=None):
    """Materialised exact sqrt-GGN factors of Sequential(Linear, ReLU, Linear) + CrossEntropy(mean).

    Returns a list of [n, P_local] matrices (n = C*batch, class-major rows), one per parameter
    (slice): exactly what BackPACK's SqrtGGNExact stores in ``param.sqrt_ggn_exact`` viewed 2-D.
    ``shard=(r, R)``: rank r keeps the r-th of R slices of the first layer's output units for the
    big first-layer weight; the small parameters go to rank 0.  ``samples=(lo, hi)``: only the rows of the
    samples lo..hi-1 of the batch (a data-parallel rank's shard: ``[C * (hi - lo), P]``, class-major over the
    shard; the forward pass still sees the whole batch, so the factors equal the corresponding rows of the full
    ones).  Synthetic-input generator, outside the timed region (torch ops).
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
        if samples is not None:
            lo_s, hi_s = samples
            S, a1, z1, X = S[:, lo_s:hi_s].contiguous(), a1[lo_s:hi_s], z1[lo_s:hi_s], X[lo_s:hi_s]
            batch = hi_s - lo_s
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


def verify_gram(facs, G, num=128, seed=0, row_index=None):
    """Check a Gram matrix built by the HIP path against fp64 dot products of sampled rows/columns.

    Checker only (torch fp64 ops on the device): ``num`` x ``num`` entries whose rows and columns cover every tile
    class of the 256-tile SYRK (first/last rows of tiles and of 16x16-tile super-blocks, the matrix corners, random
    interior points; rows > columns hit computed lower tiles, rows < columns the mirrored stores), exact symmetry of
    the whole matrix, and the trace against the fp64 squared row norms.  Returns a dict of the measured errors,
    each relative to ``sqrt(G_ii G_jj)`` (Cauchy-Schwarz scale of an entry).  ``row_index`` (data-parallel runs):
    ``facs`` hold only the rows ``row_index`` of the full factors (this rank's batch shard); entries, diagonal and
    trace are then checked on that sub-matrix of ``G``."""
    n = G.shape[0]
    m = facs[0].shape[0]
    dev = G.device
    g = torch.Generator().manual_seed(seed)
    fixed = [0, 1, 255, 256, 257, 4095, 4096, 4097, m // 2 - 1, m // 2, m - 257, m - 256, m - 2, m - 1]
    fixed = [i for i in fixed if 0 <= i < m]

    def pick(k):
        extra = torch.randint(0, m, (max(k - len(fixed), 0),), generator=g).tolist()
        return torch.tensor(sorted(set(fixed + extra)), device=dev)

    I, J = pick(num), pick(num)
    ref = torch.zeros((I.numel(), J.numel()), dtype=torch.float64, device=dev)
    sq = torch.zeros(m, dtype=torch.float64, device=dev)
    for A in facs:
        ref += A[I].double() @ A[J].double().T
        for lo in range(0, m, 8192):  # fp64 squared row norms in slabs (bounded checker memory)
            sq[lo:lo + 8192] += (A[lo:lo + 8192].double() ** 2).sum(1)
    rows = torch.arange(n, device=dev) if row_index is None else row_index.to(dev)
    got = G[rows[I]][:, rows[J]].double()
    scale = torch.sqrt(sq[I])[:, None] * torch.sqrt(sq[J])[None, :]
    entry_err = ((got - ref).abs() / scale).max().item()
    sym = all(torch.equal(G[lo:lo + 4096], G[:, lo:lo + 4096].T) for lo in range(0, n, 4096))
    diag = G.diagonal().double()[rows]
    diag_err = ((diag - sq).abs() / sq.clamp_min(1e-300)).max().item()
    trace_err = abs(diag.sum().item() - sq.sum().item()) / sq.sum().item()
    return {"entries": int(I.numel() * J.numel()), "entry_err": entry_err, "symmetric": bool(sym),
            "diag_err": diag_err, "trace_err": trace_err, "trace": sq.sum().item()}


def verify_symeig(G, w, Z, block=4096):
    """Size-independent properties of an eigendecomposition ``G = Z diag(w) Z^T`` (checker: torch fp64 matmuls on the
    device, in slabs): ascending order, trace and Frobenius identities, and -- over ALL eigenvectors, every product
    accumulated in fp64 -- the per-eigenpair 2-NORM residual ``max_i ||G z_i - w_i z_i||_2`` and the orthonormality
    ``max |Z^T Z - I|`` (the properties of test/linalg/test_eigh.py:123-144 of the reference).

    Why the 2-norm: for symmetric ``G`` and a unit vector ``z``, ``||G z - w z||_2 <= eps`` PROVES that an exact
    eigenvalue of ``G`` lies within ``eps`` of ``w``.  ``residual_2norm_fp64 <= 1e-5 lambda_max`` is therefore BASELINE's
    "eigenvalues within 1e-5 rel-err" (scoped by lambda_max as test/linalg/test_eigvalsh.py:55-60 scopes its rtol/atol),
    for every one of the n eigenpairs; an entry-wise maximum would only bound that norm up to a factor sqrt(n)."""
    n = G.shape[0]
    lam = w[-1].item()
    out = {"ascending": bool((w[1:] >= w[:-1]).all()), "lambda_max": lam}
    out["trace_err"] = abs(w.double().sum().item() - G.diagonal().double().sum().item()) / (n ** 0.5 * lam)
    fro2 = sum((G[i:i + block].double() ** 2).sum().item() for i in range(0, n, block))
    out["fro_err"] = abs((w.double() ** 2).sum().item() - fro2) / fro2
    if Z is not None:
        wd = w.double()
        res2 = res_inf = orth = 0.0
        orth_sq, norm_dev = 0.0, 0.0
        for i in range(0, n, block):
            Zi = Z[:, i:i + block].double()                      # [n, b] column slab of the eigenvectors, promoted once
            b = Zi.shape[1]
            R = torch.empty((n, b), dtype=torch.float64, device=G.device)
            for r in range(0, n, block):                         # G promoted to fp64 slab by slab
                R[r:r + block] = G[r:r + block].double() @ Zi
            R -= Zi * wd[i:i + b]
            cn = Zi.norm(dim=0)
            norm_dev = max(norm_dev, (cn - 1).abs().max().item())
            res2 = max(res2, (R.norm(dim=0) / cn).max().item())
            res_inf = max(res_inf, R.abs().max().item())
            del R
            for r in range(i, n, block):                         # Z^T Z - I, upper block triangle (it is symmetric)
                S = Z[:, r:r + block].double().T @ Zi
                if r == i:
                    S.diagonal().sub_(1.0)
                orth = max(orth, S.abs().max().item())
                orth_sq += (1.0 if r == i else 2.0) * S.pow(2).sum().item()
                del S
            del Zi
        out["residual_2norm_fp64"] = res2 / lam
        out["residual_err"] = res_inf / lam
        out["orth_err"] = orth
        out["orth_rms"] = (orth_sq / float(n) ** 2) ** 0.5
        out["norm_err"] = norm_dev
        out["orth_noise_scale_sqrt_n_eps"] = float(n) ** 0.5 * 2.0 ** -24
        out["checker"] = "fp64 accumulation over all n eigenvectors (G and Z promoted slab by slab)"
    return out


# what "verified" means (tests/test_headline_gpu.py asserts the same bounds)
# entries: fp32 contraction of length 4e5 with two accumulation levels, relative to sqrt(G_ii G_jj) (measured with
# the fp32 MFMA kernel / the bf16-pipe kernel: 1.3e-6 / 2.6e-6 off the diagonal, 2.5e-6 / 3.7e-6 on it where all terms
# are positive and rounding cannot cancel, 3.5e-7 / 1.2e-6 for the trace)
# residual_2norm_fp64: BASELINE's 1e-5 eigenvalue tolerance, literally (see verify_symeig); residual_err is the entry-wise
# maximum of the same fp64 residual (<= the 2-norm); orth_err: all of Z^T Z - I in fp64 (round 4 sampled 256 columns)
VERIFY_BOUNDS = {"entry_err": 5e-6, "diag_err": 1e-5, "trace_err": 4e-6, "eig_trace_err": 1e-5, "fro_err": 1e-4,
                 "orth_err": 2e-5, "residual_err": 1e-5, "residual_2norm_fp64": 1e-5}


def verified_ok(vg, ve):
    ok = vg["symmetric"] and vg["entry_err"] <= VERIFY_BOUNDS["entry_err"] and vg["diag_err"] <= VERIFY_BOUNDS["diag_err"]
    ok = ok and vg["trace_err"] <= VERIFY_BOUNDS["trace_err"] and ve["ascending"]
    ok = ok and ve["trace_err"] <= VERIFY_BOUNDS["eig_trace_err"] and ve["fro_err"] <= VERIFY_BOUNDS["fro_err"]
    if "orth_err" in ve:
        ok = ok and ve["orth_err"] <= VERIFY_BOUNDS["orth_err"] and ve["residual_err"] <= VERIFY_BOUNDS["residual_err"]
        ok = ok and ve["residual_2norm_fp64"] <= VERIFY_BOUNDS["residual_2norm_fp64"]
    return bool(ok)


def mlp_factorised_factors(dims, batch, device, seed=0):
    """The same MLP's factors in the form ViViTGGNExact keeps them (vivit/extensions/secondorder/vivit/linear.py:41-42):
    per Linear layer ``(s [C, N, out], z [N, in])`` with ``V_t(weight)[c,n,o,i] = s[c,n,o] z[n,i]``, ``V_t(bias) = s``."""
    d_in, d_h, C = dims
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        lin1 = torch.nn.Linear(d_in, d_h)
        lin2 = torch.nn.Linear(d_h, C)
        X = torch.rand(batch, d_in)
    W1, b1, W2, b2 = (t.detach().to(device) for t in (lin1.weight, lin1.bias, lin2.weight, lin2.bias))
    X = X.to(device)
    with torch.no_grad():
        z1 = X @ W1.T + b1
        a1 = z1.clamp_min(0)
        p = (a1 @ W2.T + b2).softmax(1)
        sq = p.sqrt()
        S = (torch.einsum("nv,vc->vnc", sq, torch.eye(C, device=device)) - torch.einsum("nv,nc->vnc", sq, p)) / math.sqrt(batch)
        M1 = (S @ W2) * (z1 > 0).unsqueeze(0)
    return [(S.contiguous(), a1.contiguous()), (M1.contiguous(), X.contiguous())]  # layer 2, layer 1


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_T0 = time.perf_counter()


def _progress(msg):
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def _median_time(fn, repeats=3, warm=True):
    if warm:
        fn()  # warm-up
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def _tune_threads(dims=(784, 512, 10), C=10, gram_batch=256, eig_n=5120):
    """The box may expose far more hardware threads than this process' CPU share (the r01 baseline ran 128 threads
    on a 16-core share and was ~5x too slow; OpenMP spin-waits under a CFS quota can be catastrophically slow): time the
    SAMPLE'S OWN two phases -- the oracle's einsum Gram of the first-layer weight at n = C * gram_batch and
    torch.linalg.eigh at n = eig_n -- for ascending thread counts up to what the affinity mask allows (32 and 64
    included where present), and keep the fastest per phase-sum; stop as soon as more threads are clearly slower."""
    from oracle import vivit_oracle as oracle

    ncpu = os.cpu_count() or 8
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = ncpu
    ncpu = min(ncpu, affinity)
    _progress(f"cpu baseline: os.cpu_count() = {os.cpu_count()}, len(os.sched_getaffinity(0)) = {affinity}")
    # (on a many-core host the 8-thread candidate has lost every probe so far and costs 15 s of the leg: start at 16 there)
    cands = [c for c in ((16, 32, 64, 128) if ncpu >= 64 else (8, 16, 32, 64, 128)) if c <= ncpu] or [ncpu]
    if ncpu not in cands and ncpu < 128:
        cands.append(ncpu)
    facs = mlp_sqrt_ggn_factors(dims, gram_batch, torch.device("cpu"))
    V = [max(facs, key=lambda f: f.shape[1]).view(C, gram_batch, -1)]           # the first-layer weight: 98.6 % of the flops
    g = torch.Generator().manual_seed(0)
    A = torch.randn(eig_n, 64, generator=g)
    S = A @ A.T + torch.eye(eig_n)
    best, best_t, table = cands[0], float("inf"), {}
    for c in cands:
        torch.set_num_threads(c)
        tg = _median_time(lambda: oracle.compute_gram_mat(V, start_dim=2, flatten=True), repeats=1, warm=c == cands[0])
        te = _median_time(lambda: torch.linalg.eigh(S), repeats=1, warm=c == cands[0])
        # weight the phases as the full problem does (Gram ~ n^2 P, eigh ~ n^3): extrapolate each to the headline size
        t = tg + te
        table[c] = {"einsum_gram_s": round(tg, 3), "eigh_s": round(te, 3)}
        _progress(f"cpu baseline: thread probe {c} threads: einsum Gram (n={C * gram_batch}) {tg:.2f} s, eigh (n={eig_n}) {te:.2f} s")
        if t < best_t * 0.97:  # prefer fewer threads unless clearly faster
            best, best_t = c, t
        elif t > best_t * 1.1:
            break              # more threads are slower on this share of the box: do not try even more (bounds the leg's run time)
    torch.set_num_threads(best)
    return best, {"affinity": affinity, "cpu_count": os.cpu_count(), "probe": table}


def cpu_baseline(dims, C, full_n, full_P, batches=(128, 256, 512), eig_batches=(512, 1024, 1600), repeats=3):
    """The oracle (CPU restatement of the reference algorithm) timed on this host on a bounded sample, both flavours
    the reference has for this MLP, never mixed:
      materialised -- einsum Gram over every parameter with the full 2 n^2 P work (vivit/utils/gram.py:230-232,
        104-116: what SqrtGGNExact + vivit.optim / GramSqrtGGNExact run), then torch.linalg.eigh, the successor of
        Tensor.symeig (vivit/linalg/eigh.py:248-250);
      factorised   -- the Linear fast path of ViViTGGNExact (vivit/extensions/secondorder/vivit/linear.py:72-75) for
        the weights + materialised biases, then the same eigh.
    Median of ``repeats`` after one warm-up; the einsum Gram is timed at ``batches`` (it is the expensive phase of the
    sample: 2 n^2 P flop), eigh/eigvalsh at the larger ``eig_batches`` (on the same MLP's Gram matrix, built with the
    factorised form) because LAPACK only approaches its n^3 regime there.  The exponent of each phase is a least-squares
    FIT over its three sizes (never steeper than the flop count; the fit's worst residual is reported) and extrapolates the
    largest sample -- n = 5120 for the Gram, n = 16 000 for eigh -- to the full n (stated in ``sample``).
    Thread count tuned first (``_tune_threads``)."""
    from oracle import vivit_oracle as oracle

    threads, table = _tune_threads(dims, C, gram_batch=sorted(batches)[min(1, len(batches) - 1)], eig_n=C * min(eig_batches))
    _progress(f"cpu baseline: {threads} threads (probe {table})")
    cpu = torch.device("cpu")

    def fact_gram_fn(fz):
        def fact_gram():
            G = None
            for s_, z_ in fz:
                Gw = oracle.linear_weight_gram(s_, z_)                       # weight: (z z^T) o (s s^T)
                Gb = oracle.pairwise_dot(s_, start_dim=2, flatten=False)     # bias: V_t = s
                G = Gw + Gb if G is None else G + Gw + Gb
            return oracle.reshape_as_square(G)

        return fact_gram

    gram_rows, eig_rows = [], []
    for bi, b in enumerate(batches):
        facs = mlp_sqrt_ggn_factors(dims, b, cpu)
        n = facs[0].shape[0]
        V = [f.view(C, b, -1) for f in facs]
        # (the warm-up run is only needed once per process: thread pool, allocator; the LARGEST size of each phase is run once
        # -- it is the expensive sample and the least noisy one)
        last = bi == len(batches) - 1 and len(batches) > 2
        g_reps = 1 if last else (repeats if bi == 0 else max(2, repeats - 1))   # (seconds per run from the second size on: two runs)
        t_gram = _median_time(lambda: oracle.compute_gram_mat(V, start_dim=2, flatten=True), g_reps, warm=bi == 0)
        del V, facs
        t_fact = _median_time(fact_gram_fn(mlp_factorised_factors(dims, b, cpu)), repeats)
        _progress(f"cpu baseline: batch {b} (n={n}) einsum Gram {t_gram:.2f} s, factorised Gram {t_fact:.3f} s")
        gram_rows.append({"batch": b, "n": n, "gram_materialised_s": t_gram, "gram_factorised_s": t_fact, "repeats": g_reps})
    fact_rows = []
    for bi, b in enumerate(eig_batches):
        # the eigh input IS the factorised Gram: time its construction here as well -- the factorised line is then fitted through
        # n = 5120 / 10 240 / 20 480, all out of cache like the full size (the small sizes above run in L2/L3: their times jump
        # 20x from n = 2560 to 5120 and no single exponent fits them -- fit residual 1.6 in round 5)
        fg = fact_gram_fn(mlp_factorised_factors(dims, b, cpu))
        t0 = time.perf_counter()
        gram = fg()
        t_fg = time.perf_counter() - t0
        if bi == 0:   # first call of the leg at this size: once more, warm
            t0 = time.perf_counter()
            gram = fg()
            t_fg = time.perf_counter() - t0
        fact_rows.append({"batch": b, "n": gram.shape[0], "gram_factorised_s": t_fg, "repeats": 1})
        n = gram.shape[0]
        last = bi == len(eig_batches) - 1 and len(eig_batches) > 2
        reps = 1 if (last or (bi == len(eig_batches) - 2 and len(eig_batches) > 2)) else repeats
        t_eig = _median_time(lambda: oracle.tensor_symeig(gram, eigenvectors=True), reps, warm=bi == 0)
        row = {"batch": b, "n": n, "eigh_s": t_eig, "repeats": reps}
        if not last:   # (values only is a secondary line: its two smaller sizes are enough)
            row["eigvalsh_s"] = _median_time(lambda: oracle.tensor_symeig(gram, eigenvectors=False), reps, warm=bi == 0)
        _progress(f"cpu baseline: batch {b} (n={n}) eigh {t_eig:.2f} s" + (f", eigvalsh {row['eigvalsh_s']:.2f} s" if not last else ""))
        eig_rows.append(row)
        del gram

    def extrap(rows, key, textbook):
        """Least-squares line through (log n, log t) of all sampled sizes, slope capped at the flop count's exponent;
        the prediction at full_n and the fit's worst residual |log(t_measured / t_fit)| (0 for two points)."""
        pts = [(math.log(r["n"]), math.log(r[key])) for r in rows if key in r and r[key] > 0]
        if len(pts) < 2:
            r = rows[-1]
            return r[key] * (full_n / r["n"]) ** textbook, textbook, 0.0
        mx, my = sum(x for x, _ in pts) / len(pts), sum(y for _, y in pts) / len(pts)
        sxx = sum((x - mx) ** 2 for x, _ in pts)
        e = min(sum((x - mx) * (y - my) for x, y in pts) / sxx, textbook)   # never steeper than the flop count
        # anchor the (possibly capped) line at the LARGEST sample: the regime closest to the full size
        xl, yl = pts[-1]
        resid = max(abs(y - (yl + e * (x - xl))) for x, y in pts)
        return math.exp(yl + e * (math.log(full_n) - xl)), e, resid

    tg, eg, rg = extrap(gram_rows, "gram_materialised_s", 2.0)
    tf, ef, rf = extrap(fact_rows if len(fact_rows) >= 2 else gram_rows, "gram_factorised_s", 2.0)
    te, ee, re_ = extrap(eig_rows, "eigh_s", 3.0)
    tv, ev_, rv = extrap(eig_rows, "eigvalsh_s", 3.0)
    value = full_n / (tg + te)
    return {
        "value": value,
        "unit": "eigenpairs/s",
        "cores": threads,
        "kind": "port",
        "cpu_model": _cpu_model(),
        "thread_probe_s": table,
        "samples": {"gram": gram_rows, "eigh": eig_rows, "gram_factorised_large": fact_rows},
        "fitted_exponents": {"gram_materialised": eg, "gram_factorised": ef, "eigh": ee, "eigvalsh": ev_},
        "fit_points": {"gram": len(gram_rows), "eigh": len(eig_rows), "gram_factorised": len(fact_rows) if len(fact_rows) >= 2 else len(gram_rows)},
        "fit_worst_log_residual": {"gram_materialised": rg, "gram_factorised": rf, "eigh": re_, "eigvalsh": rv},
        "extrapolated_s": {"gram_materialised": tg, "gram_factorised": tf, "eigh": te, "eigvalsh": tv},
        "materialised_eigenpairs_per_s": value,
        "factorised_eigenpairs_per_s": full_n / (tf + te),
        "sample": (
            f"same MLP (P={full_P}), median of {repeats} after a warm-up (largest size of each phase: one run), {threads} threads "
            f"({_cpu_model()}): einsum Gram at n={[r['n'] for r in gram_rows]}, torch.linalg.eigh at n={[r['n'] for r in eig_rows]}; "
            f"value = materialised line extrapolated to n={full_n} along the least-squares exponents of the {len(gram_rows)} / "
            f"{len(eig_rows)} sizes (Gram n^{eg:.2f}, eigh n^{ee:.2f}; worst fit residual {max(rg, re_) * 100:.1f} %), anchored at the largest sample"
        ),
    }


def _spawn_ranks(nranks):
    """`python bench.py --gpus N` launched bare: start the N ranks as child processes BEFORE this process touches
    the GPU (never exec from a process that initialised HIP), one rank per GPU, and forward rank 0's JSON line."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (own process group per child: the watchdog below can end a rank together with anything it started)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True))
    return _watch_ranks(procs)


def _watch_ranks(procs, poll_s=0.5, grace_s=20.0):
    """Watchdog of the self-spawned ranks: poll ALL children; as soon as one exits non-zero (out of memory, RCCL
    initialisation, an assertion) the others -- who would sit in a collective until the driver's timeout -- are
    terminated (SIGTERM to their process groups, SIGKILL after ``grace_s``) and the launcher exits non-zero.  Only the
    fresh children are ever signalled, by the PIDs this launcher created."""
    import signal

    def end(p, sig):
        if p.poll() is None:
            try:
                os.killpg(p.pid, sig)
            except (ProcessLookupError, PermissionError):
                pass

    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad and failed is None:
            failed = bad[0]
            print(f"[bench] rank {failed[0]} exited with code {failed[1]}: terminating the other ranks", file=sys.stderr, flush=True)
            for p in procs:
                end(p, signal.SIGTERM)
            deadline = time.monotonic() + grace_s
        if all(c is not None for c in codes):
            break
        if failed is not None and time.monotonic() > deadline:
            for p in procs:
                end(p, signal.SIGKILL)
        time.sleep(poll_s)
    if failed is not None:
        return abs(failed[1]) or 1
    return 0


PANEL_EXCHANGE_US = 2.04        # one-XCD barrier + exchange (profiles/r03_grid_barrier_probe.log, mode 2)
PANEL_LOCAL_LAUNCHES = 15       # dependent small launches per panel besides the QR (profiles/r06_bench_n40960_kernel_stats.csv)
DEPENDENT_LAUNCH_US = 3.0       # empty dependent kernel boundary (scripts/probe/launch_rate.hip: 3.0-3.3 us)
VALU_F64_PEAK_TF = 78.6         # fp64 vector peak (MI355X_MICROARCH.md: 256 CUs x 128 flop/clk x 2.4 GHz)


def _stage_rooflines(stage_ms, n, steps, vectors, row_frac=1.0, split=0):
    """Per-stage roofline of the eigensolver from the library's stage marks (ms summed over the timed steps).
    Algorithmic work per stage: band reduction (4/3) n^3 flop on MFMA; bulge chasing: n^2/(2 nb)
    tasks that each read and write two nb x nb blocks; the two back-transformations 2 n^3 flop each on MFMA."""
    names = {1: "prepare (scale scan + mirror)", 2: "sy2sb (full -> band): panel QR + panel-local products",
             9: "sy2sb: streaming panel products P^T = V^T A22",
             10: "sy2sb: delayed rank-1024 trailing updates (bf16 pipe)",
             3: "sb2st (band -> tridiagonal, bulge chasing)",
             4: "tridiagonal eigenproblem (divide & conquer | multisection)", 5: "Q2 back-transformation",
             6: "Q1 back-transformation", 7: "sort + transpose into the output", 8: "sytrd (one-stage tridiagonalisation)"}
    # (csrc/q2slide.hip:q2_slide_ok: the sliding-window kernel takes over when this many eigenvector rows are transformed)
    q2_bf16 = (bool(split) and n * row_frac >= int(os.environ.get("VIVIT_Q2_SLIDE_MIN_ROWS", "14336"))
               and os.environ.get("VIVIT_Q2_SLIDE", "1") != "0" and n % 4 == 0)
    # (csrc/gemm_f32.hip:gemm64_bx_enabled: the 64-row streaming product on the bf16 pipe, then bound by the HBM stream)
    g64_bf16 = bool(split) and os.environ.get("VIVIT_GEMM64_BX", "1") != "0"
    nb = 64
    n3 = float(n) ** 3
    work = {
        1: ("hbm", 4.0 * n * n * 1.5, "B"),
        # A LATENCY model, not a throughput one: the panel QR is a chain of 64 column steps per panel, each ending in an exchange
        # between the 32 workgroups of one XCD (2.04 us measured for barrier + exchange on one XCD: profiles/r03_grid_barrier_probe.log,
        # mode 2); around it a panel costs PANEL_LOCAL_LAUNCHES dependent small launches (T factor, coefficient products on 64-wide
        # operands, load / store of the panel; count from profiles/r06_bench_n40960_kernel_stats.csv) at 3.0 us per dependent
        # kernel boundary (scripts/probe/launch_rate.hip).  No look-ahead can shorten the chain: in a TWO-sided reduction panel
        # p + 1's columns need W_p, i.e. the complete streaming product A22 V_p (stage 9) of panel p.
        2: ("latency", (n / nb - 1) * (nb * PANEL_EXCHANGE_US + PANEL_LOCAL_LAUNCHES * DEPENDENT_LAUNCH_US) * 1e-6, "s"),
        # every panel reads its trailing matrix once: 4 B x sum_p m_p^2 = n^3 / 48 bytes, 2 x 64 flop per element
        9: ("hbm", n3 / 48.0, "B") if g64_bf16 else ("mfma", 2.0 / 3.0 * n3, "flop"),
        10: ("mfma", 2.0 / 3.0 * n3, "flop"),
        3: ("hbm", (n * n / (2.0 * nb)) * 4 * nb * nb * 4.0, "B"),
        # divide & conquer: the merges' cost depends on deflation (spectrum); what ANY with-vectors tridiagonal solver must move is
        # the n x n eigenvector matrix, written once and read once by the final rank permutation: the floor stated here.  Values
        # only (Sturm multisection): n eigenvalues x 9 rounds x 8 shifts x n recurrence steps of ~6 fp64 operations on the vector pipe
        4: ("hbm", 8.0 * n * n, "B") if vectors else ("valu64", 9.0 * 8.0 * 6.0 * n * n, "flop"),
        5: ("mfma", 2.0 * n3 * row_frac, "flop"),   # back-transformations: only this rank's eigenvector rows
        6: ("mfma", 2.0 * n3 * row_frac, "flop"),
        7: ("hbm", 8.0 * n * n, "B"),
        8: ("hbm", 2.0 / 3.0 * n3, "B"),
    }
    out = []
    for k in (1, 2, 9, 10, 3, 4, 5, 6, 7, 8):
        sec = stage_ms[k] / 1e3 / max(steps, 1)
        if sec <= 0:
            continue
        bound, amount, unit = work[k]
        row = {"stage": names[k], "seconds": sec, "bound": bound}
        if bound is None:
            out.append(row)
            continue
        if k == 4:
            row["note"] = ("floor of any with-vectors tridiagonal solver (write Z once, read it once for the final permutation); the merges' own "
                           "traffic depends on deflation" if vectors else "Sturm multisection, fp64 recurrences on the vector pipe")
        if bound == "latency":
            row.update({"model_seconds": amount, "frac": amount / sec, "unit": "s",
                        "model": f"(n/64 - 1) panels x (64 exchanges x {PANEL_EXCHANGE_US} us + {PANEL_LOCAL_LAUNCHES} dependent launches x "
                                 f"{DEPENDENT_LAUNCH_US} us); frac = model / measured"})
        elif bound == "valu64":
            ach = amount / sec / 1e12
            row.update({"flops": amount, "achieved": ach, "peak": VALU_F64_PEAK_TF, "unit": "TFLOP/s", "frac": ach / VALU_F64_PEAK_TF})
        elif bound == "mfma":
            ach = amount / sec / 1e12
            # each stage against the pipe it runs on: Q1's products, the band reduction's trailing updates and the
            # sliding-window Q2 kernel form fp32 products from exact bf16 splits (roofline = bf16 peak / split); the band
            # reduction's streaming panel product (VIVIT_GEMM64_BX=0) and the block-step Q2 kernels are fp32 MFMA kernels
            on_bf16 = split and (k in (6, 10) or (k == 5 and q2_bf16))
            peak = MFMA_BF16_PEAK_TF / split if on_bf16 else MFMA_F32_PEAK_TF
            row["pipe"] = f"bf16 MFMA, {split} partial products per fp32 product" if on_bf16 else "fp32 MFMA"
            row.update({"flops": amount, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak})
        elif bound == "hbm":
            ach = amount / sec / 1e9
            row.update({"bytes": amount, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS})
            if k == 9:   # the same seconds against the matrix pipe it runs on
                tf = 2.0 / 3.0 * n3 / sec / 1e12
                row.update({"pipe": f"bf16 MFMA, {split} partial products per fp32 product", "flops": 2.0 / 3.0 * n3,
                            "mfma_tflops": tf, "mfma_frac": tf / (MFMA_BF16_PEAK_TF / split)})
        out.append(row)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mlp784-512-10_b4096", choices=sorted(WORKLOADS))
    ap.add_argument("--values-only", action="store_true", help="EigvalshComputation flavour (no eigenvectors)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the parity check after the timed loop")
    ap.add_argument("--no-secondary", action="store_true", help="skip the values-only / top-10 secondary lines")
    ap.add_argument("--cpu-batches", default="128,256,512", help="batch sizes of the CPU baseline's einsum-Gram sample")
    ap.add_argument("--cpu-eig-batches", default="512,1024,1600", help="batch sizes of the CPU baseline's eigh sample (n = 10 x batch; 2048 = "
                    "half the full size costs 62 s for its one eigh and put the driver's run at 381 s; 1600: 31 s)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations (configs block)")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="torch.distributed backend of a multi-rank run (default nccl = RCCL over xGMI; gloo: functional "
                         "runs, several ranks may share one card)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP device); there is no CPU fallback")
    # (functional testing on a 1-GPU box: VIVIT_DIST_BACKEND=gloo lets several ranks share device 0)
    backend = args.backend or os.environ.get("VIVIT_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    if world > torch.cuda.device_count():
        # ranks share a card (functional runs only): the one-XCD persistent kernels want XCD 0 of their GPU to themselves
        os.environ.setdefault("VIVIT_SYTRD_PERSIST", "0")
        os.environ.setdefault("VIVIT_QR_PERSIST", "0")
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

    import ctypes

    from vivit_amd import _lib, kernels
    from vivit_amd import distributed as vdist

    # what the process group REALLY is (not argv): world size, backend string, every rank's device -- into the JSON line
    mine = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "device": torch.cuda.get_device_name(dev_index),
            "gcn_arch": getattr(torch.cuda.get_device_properties(dev_index), "gcnArchName", None),
            "cus": torch.cuda.get_device_properties(dev_index).multi_processor_count, "pid": os.getpid()}
    if dist is not None:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
        world_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": ranks_info,
                      "distinct_devices": len({(r["device_index"]) for r in ranks_info})}
        if world_info["world_size"] != world:
            raise SystemExit(f"process group has {world_info['world_size']} ranks, --gpus says {world}")
    else:
        world_info = {"world_size": 1, "backend": None, "ranks": [mine], "distinct_devices": 1}

    dims, batch, C = WORKLOADS[args.workload]
    n = C * batch
    P_total = dims[0] * dims[1] + dims[1] + dims[1] * dims[2] + dims[2]
    if batch % world != 0:
        raise SystemExit(f"batch {batch} is not divisible by {world} ranks")
    Ng = batch // world
    # data parallel: this rank's batch shard of every factor, [C * Ng, P_p] (class-major over the shard)
    facs = mlp_sqrt_ggn_factors(dims, batch, device, samples=(rank * Ng, (rank + 1) * Ng) if world > 1 else None)
    G = torch.empty((n, n), dtype=torch.float32, device=device) if world == 1 else None
    lib = _lib.load()
    vectors = not args.values_only

    ev = {k: [torch.cuda.Event(enable_timing=True) for _ in range(5)] for k in range(args.steps + args.warmup + 1)}

    def build_gram(e=None):
        if world == 1:
            for k, A in enumerate(facs):
                kernels.gram_syrk(A, out=G, alpha=1.0, beta=0.0 if k == 0 else 1.0)
            if e is not None:
                e[1].record(), e[2].record()
            return G
        # batch shards -> parameter shards (all-to-all in column chunks, chunk j + 1 in flight while the SYRK of chunk j
        # runs) -> full-size partial SYRKs -> all-reduce of the packed lower triangle (vivit_amd/distributed.py)
        acc = vdist.BatchShardedGram(C, Ng)
        if e is not None:
            e[1].record()
        for A in facs:
            acc.add_factor(A.view(C, Ng, -1))
        if e is not None:
            e[2].record()
        return acc.finalize().view(n, n)

    def step(idx):
        e = ev[idx]
        e[0].record()
        Gm = build_gram(e)
        e[3].record()
        if world > 1 and vectors:
            # reduction + tridiagonal solve replicated, back-transformations sharded by eigenvector, all-gather
            w, Z = vdist.symeig(Gm, overwrite=True)
        else:
            w, Z = kernels.symeig(Gm, eigenvectors=vectors, overwrite=True)
        e[4].record()
        return w, Z

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        _progress(f"factors resident ({sum(f.numel() for f in facs) * 4 / 1e9:.1f} GB on rank 0); warm-up x{args.warmup}")
    for i in range(args.warmup):
        w, Z = step(i)
        if rank == 0:
            _progress(f"warm-up step {i + 1} issued")
    if dist is not None and args.warmup > 0:
        # self-check before timing: the replicated stages (band reduction, bulge chase, tridiagonal solve) run on every rank
        # and nothing is broadcast -- the eigenvalues (and the gathered eigenvectors) must be BIT-identical on all ranks
        gathered = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(gathered, w)
        same = all(torch.equal(gathered[0], g_) for g_ in gathered[1:])
        if same and Z is not None:
            chk = Z.view(torch.int32).to(torch.int64).sum().reshape(1)      # order-independent checksum of the bit patterns
            sums = [torch.empty_like(chk) for _ in range(world)]
            dist.all_gather(sums, chk)
            same = all(torch.equal(sums[0], s_) for s_ in sums[1:])
        if not same:
            raise SystemExit("ranks disagree bitwise on the eigendecomposition: the replicated stages are not deterministic")
        if rank == 0:
            _progress(f"self-check: eigenvalues and eigenvector checksum bit-identical on all {world} ranks")
        del gathered
    barrier()
    if rank == 0:
        _progress(f"timing {args.steps} steps")
    lib.vivit_profile_begin(64)
    t0 = time.perf_counter()
    for i in range(args.steps):
        w, Z = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    stage_ms = (ctypes.c_double * 16)()
    lib.vivit_profile_stages(stage_ms, 16)
    prof = (ctypes.c_double * 6)()
    lib.vivit_profile_end(prof)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    def phase(a, b):
        return sum(ev[args.warmup + i][a].elapsed_time(ev[args.warmup + i][b]) for i in range(args.steps)) / 1e3 / args.steps

    # run-to-run spread of THIS rank's steps (device time between the first and last event of each timed step): lets a reader
    # tell box-to-box differences (+- 2 % between leases) from noise inside one run
    per_step = sorted(ev[args.warmup + i][0].elapsed_time(ev[args.warmup + i][4]) for i in range(args.steps))
    step_spread = {"min": per_step[0], "median": per_step[len(per_step) // 2], "max": per_step[-1], "n": len(per_step),
                   "clock": "HIP events around each step on the launch stream (rank 0)"}

    if world == 1:
        exch_s, gram_s, ar_s = 0.0, phase(0, 1), 0.0
    else:
        # the exchange is pipelined behind the SYRKs: phase (1, 2) is exchange + Gram, (0, 1) the shard-size handshake
        exch_s, gram_s, ar_s = phase(0, 1), phase(1, 2), phase(2, 3)
    eig_s = phase(3, 4)

    if rank == 0:
        _progress(f"timed region done: {elapsed / args.steps:.3f} s per step")
    verified = None
    if not args.no_verify:
        # parity of THIS shape (not timed): rebuild G, sampled fp64 entries / symmetry / trace, then the
        # eigendecomposition's properties on that matrix
        del w, Z
        torch.cuda.empty_cache()
        Gv = build_gram()
        row_index = None
        if world > 1:
            row_index = (torch.arange(C, device=device)[:, None] * batch + rank * Ng + torch.arange(Ng, device=device)[None, :]).reshape(-1)
        vg = verify_gram(facs, Gv, num=128, row_index=row_index)
        if rank == 0:
            _progress(f"verify: Gram entries ok={vg['entry_err'] <= VERIFY_BOUNDS['entry_err']} ({vg['entry_err']:.2e})")
        wv, Zv = (vdist.symeig(Gv) if (world > 1 and vectors) else kernels.symeig(Gv, eigenvectors=vectors))
        torch.cuda.synchronize()
        if rank == 0:
            _progress("verify: eigendecomposition of the rebuilt Gram matrix done, checking its properties")
        ve = verify_symeig(Gv, wv, Zv)
        if rank == 0:
            _progress(f"verify: fp64 2-norm residual {ve.get('residual_2norm_fp64', float('nan')):.2e} lambda_max, "
                      f"orthonormality {ve.get('orth_err', float('nan')):.2e}")
        ve["trace_err_eig"] = ve.pop("trace_err")
        verified = {"ok": verified_ok(vg, dict(ve, trace_err=ve["trace_err_eig"])), "gram": vg, "symeig": ve,
                    "bounds": VERIFY_BOUNDS,
                    "how": "sampled Gram entries (all tile classes) vs fp64 dot products, exact symmetry, trace; ascending "
                           "order, trace/Frobenius identities; over ALL eigenvectors with fp64 accumulation: per-eigenpair 2-norm "
                           "residual max_i ||G z_i - w_i z_i||_2 / lambda_max (bounds every eigenvalue's error) and max |Z^T Z - I|"}
        del Gv, wv, Zv

    secondary_lines = None
    if world == 1 and not args.no_secondary:
        # secondary lines: the same Gram matrix, (i) values only (EigvalshComputation flavour),
        # (ii) criterion = top-10 (reduction + all eigenvalues, host callback, 10 eigenvectors)
        def timed(fn):
            Gs = build_gram()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out_ = fn(Gs)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / 1e3, out_

        t_vals, (w_only, _) = timed(lambda Gs: kernels.symeig(Gs, eigenvectors=False, overwrite=True))

        def top10(Gs):
            plan = kernels.symeig_reduce(Gs, overwrite=True)
            keep = list(range(n - 10, n))            # criterion callback on the host
            return plan.evals, plan.select(keep)

        t_top, (w_top, Z_top) = timed(top10)
        Gs = build_gram()
        res = (Gs @ Z_top - Z_top * w_top[-10:]).abs().max().item() / w_top[-1].item()
        orth = (Z_top.T @ Z_top - torch.eye(10, device=device)).abs().max().item()
        secondary_lines = {
            "eigvalsh_s": t_vals, "eigenvalues_per_s_incl_gram": n / (gram_s + t_vals),
            "eigh_top10_s": t_top, "top10_residual_err": res, "top10_orth_err": orth,
            "top10_minus_eigvalsh_s": t_top - t_vals,
            "note": "same Gram matrix; top-10 = vivit_symeig_reduce_f32 + host criterion + vivit_symeig_select_f32 (K = 10)",
        }
        del Gs, Z_top
        if rank == 0:
            _progress(f"secondary: eigvalsh {t_vals:.2f} s, eigh top-10 {t_top:.2f} s (residual {res:.1e}, orth {orth:.1e})")
        # (ii-b) the FACTORISED Gram build of the same MLP (what ViViTGGNExact does for Linear weights,
        # vivit/extensions/secondorder/vivit/linear.py:72-75: (z z^T) o (s s^T) per layer + the bias Grams s s^T) -- the
        # like-for-like partner of cpu_baseline.factorised_eigenpairs_per_s; never mixed with the materialised line
        fz = mlp_factorised_factors(dims, batch, device)
        Gf = torch.empty((n, n), dtype=torch.float32, device=device)

        def fact_gram():
            first = True
            for s_, z_ in fz:
                Gz = kernels.gram_syrk(z_)
                Gs = kernels.gram_syrk(s_.reshape(n, -1))
                kernels.gram_hadamard(Gz, Gs, C, batch, out=Gf, alpha=1.0, beta=0.0 if first else 1.0)   # weight
                Gf.add_(Gs)                                                                         # bias: V_t = s
                first = False
            return Gf

        fact_gram()
        torch.cuda.synchronize()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        fact_gram()
        b_.record()
        torch.cuda.synchronize()
        t_fact = a_.elapsed_time(b_) / 1e3
        Gm = build_gram()
        fact_err = float((Gf - Gm).abs().max() / Gm.abs().max())
        secondary_lines.update({
            "factorised_gram_s": t_fact, "factorised_vs_materialised_gram_err": fact_err,
            "factorised_eigenpairs_per_s": n / (t_fact + eig_s),
            "factorised_note": "Gram by the Linear fast path (2 SYRKs + fused Hadamard per layer, bias Grams) + the same symeig",
        })
        del Gf, Gm, fz
        if rank == 0:
            _progress(f"secondary: factorised Gram {t_fact * 1e3:.1f} ms (max deviation from the materialised Gram {fact_err:.1e})")
        # (iii) the same through the PUBLIC API: EighComputation's extension hook on the four parameters' materialised
        # factors (hook scheduling, Gram accumulation over the parameters, criterion callback on the host, back-projection
        # V e of the kept directions -- a 66.7 GB stream -- and normalisation are all inside the timed region)
        try:
            import vivit_amd
            from vivit_amd.backend.extensions import _materialised_closures

            shapes = [(dims[2], dims[1]), (dims[2],), (dims[1], dims[0]), (dims[1],)]     # W2, b2, W1, b1 (order of facs)
            prm = [torch.nn.Parameter(torch.empty(sh, device=device)) for sh in shapes]
            comp = vivit_amd.EighComputation()
            for p_, f_ in zip(prm, facs):
                setattr(p_, comp._savefield, _materialised_closures(f_.view(C, batch, *p_.shape)))
            holder = torch.nn.Module()
            for i_, p_ in enumerate(prm):
                holder.register_parameter(f"p{i_}", p_)
            holder.input0 = torch.empty(batch, 1, device=device)
            group = {"params": prm, "criterion": lambda ev: list(range(ev.numel() - 10, ev.numel()))}
            hook = comp.get_extension_hook([group])
            torch.cuda.synchronize()
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            hook(holder)
            b_.record()
            torch.cuda.synchronize()
            ev_api, vec_api = comp.get_result(group)
            t_api = a_.elapsed_time(b_) / 1e3
            norm2 = sum((v_.reshape(10, -1).double() ** 2).sum(1) for v_ in vec_api)
            secondary_lines.update({
                "api_eigh_top10_s": t_api,
                "api_eigh_top10_evals_err": float((ev_api - w_top[-10:]).abs().max() / w_top[-1]),
                "api_eigh_top10_norm_err": float((norm2 - 1).abs().max()),
                "api_note": "vivit_amd.EighComputation extension hook (Gram of 4 parameters + two-launch eigensolver + "
                            "back-projection to parameter space + normalisation), criterion = top-10",
            })
            if rank == 0:
                _progress(f"secondary: public API EighComputation top-10 {t_api:.2f} s")
            del vec_api, prm, holder, comp
        except Exception as exc:  # the secondary line must never take the headline down
            secondary_lines["api_error"] = repr(exc)

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
            split = int(lib.vivit_gemm_split_mode())
            if split == 0:
                roofline = {
                    "kernel": "gemm256_kernel<LAY_K,LAY_K> (Gram SYRK, fp32 MFMA)",
                    "bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": achieved / MFMA_F32_PEAK_TF,
                    "traffic": SYRK_TRAFFIC_BYTES_PMC.get((args.workload, world)),
                    "traffic_note": "bytes of the first-layer weight's SYRK launch (98.6 % of the Gram flops, 4.73 s), separate "
                                    "rocprofv3 --pmc passes (profiles/r01_pmc), FETCH_SIZE includes Infinity-Cache hits",
                }
            else:
                # fp32 products on the bf16 pipe: `split` bf16 MFMAs (exact partial products of the three-way operand
                # split) per fp32 multiply-add, so the pipe's roofline for ALGORITHMIC fp32 flops is its peak / split
                peak = MFMA_BF16_PEAK_TF / split
                roofline = {
                    "kernel": f"gemm256_bx_kernel<{split}, asm K loop> (Gram SYRK: fp32 operands split exactly into 3 bf16 pieces, {split} of 9 "
                              f"partial products on {'v_mfma_f32_16x16x32_bf16, two fused per instruction' if split == 6 else 'v_mfma_f32_32x32x16_bf16'}, "
                              f"fp32 accumulation) + bx_split_kernel",
                    "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "peak_note": f"dense bf16 MFMA peak {MFMA_BF16_PEAK_TF} TFLOP/s / {split} partial products per fp32 product; "
                                 f"achieved = algorithmic fp32 flops n(n+1)P per second",
                    "issued_bf16_tflops": achieved * split,
                    "vs_fp32_mfma_peak": achieved / MFMA_F32_PEAK_TF,
                    "traffic": SYRK_BX_TRAFFIC_BYTES_PMC.get((args.workload, world)) if split == 6 else None,
                    "traffic_note": "bytes of the first-layer weight's SYRK (98.6 % of the Gram flops: 98 split + 98 product launches of "
                                    "4096 columns) on N(0,1) data, separate rocprofv3 --pmc passes of the final code (profiles/r06_pmc_s16/pmc_syrk_*: "
                                    "FETCH_SIZE 2.4568e9 + 3.2238e7 KiB, WRITE_SIZE 6.3943e8 + 9.6339e7 KiB), FETCH_SIZE includes Infinity-Cache hits",
                    "clock_ghz": SYRK_BX_CLOCK_GHZ.get((args.workload, world)) if split == 6 else None,
                    "clock_note": "the chip lowers its clock under the bf16 MFMA load: frac = (matrix-pipe busy 0.82) x (clock / 2.4 GHz); NOT "
                                  "measured in this run: in-kernel stamps of a diagnostic build around the K loop (profiles/r06_bx16_timeline.log), "
                                  "GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES of a PMC pass over the 98 launches of the first-layer SYRK on N(0,1) data and on "
                                  "the bench's own factor (files: pmc_files; summary: profiles/r06_pmc_s16_summary.txt)",
                    "pmc_files": PMC_FILES if split == 6 else None,
                    "bare_loop_ceiling": BX_BARE_LOOP_CEILING if split == 6 else None,
                    "frac_of_bare_loop_on_like_data": (achieved / BX_BARE_LOOP_CEILING["half_zeros_like_the_bench_factors"]["tflops_fp32_equiv"])
                    if split == 6 else None,
                }
            roofline.update({
                "launches_sampled": int(syrk_cnt), "avg_launch_ms": syrk_ms / max(syrk_cnt, 1),
                "est_share_of_step": syrk_total_s / (elapsed / args.steps),
            })
        gram_peak = roofline["peak"] if roofline["bound"] == "mfma" else MFMA_F32_PEAK_TF
        phases_roof = [{"stage": "Gram build (SYRK over all parameters)", "seconds": syrk_total_s, "bound": "mfma",
                        "flops": syrk_flops / max(args.steps, 1), "achieved": roofline["achieved"] if roofline["bound"] == "mfma" else
                        (syrk_flops / (syrk_ms / 1e3) / 1e12 if syrk_ms > 0 else None), "peak": gram_peak,
                        "unit": "TFLOP/s"}]
        if phases_roof[0]["achieved"]:
            phases_roof[0]["frac"] = phases_roof[0]["achieved"] / gram_peak
        phases_roof += _stage_rooflines(list(stage_ms), n, args.steps, vectors, row_frac=1.0 / world,
                                        split=int(lib.vivit_gemm_split_mode()))
        secondary = {
            "gram_syrk_tflops": (syrk_flops / (syrk_ms / 1e3) / 1e12) if syrk_ms > 0 else None,
            "symv_gbs": (symv_bytes / (symv_ms / 1e3) / 1e9) if symv_ms > 0 else None,
        }
        out = {
            "metric": f"GGN eigenpairs/sec (Gram build + symeig), MLP {dims[0]}-{dims[1]}-{dims[2]}, batch={batch}",
            "value": value,
            "unit": "eigenpairs/s",
            "n_gpus": world_info["distinct_devices"],   # GPUs, not ranks: two gloo ranks on one card are ONE GPU (world.world_size says 2)
            "world": dict(world_info, eigensolver=(
                {"band_reduction": "sharded by block rows (vivit_amd.distributed.sy2sb_sharded_)", "collectives_per_solve": dict(vdist.LAST_SHARDED_COLLECTIVES),
                 "note": "all_gather: one per panel on the critical path; broadcast: the next panel's block row, asynchronous; NOT measured on more "
                         "than one GPU by the builder (no multi-GPU node): scaling is unmeasured"}
                if vdist.LAST_SHARDED_COLLECTIVES and world > 1 else
                {"band_reduction": "replicated" if world > 1 else "single GPU", "collectives_per_solve": {"all_gather": 1} if (world > 1 and vectors) else {}})),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_spread": step_spread,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "arithmetic": ("fp32 in, fp32 accumulate, fp32 out; the large Gram products form every fp32 product from the exact "
                           "three-way bf16 split of its operands (6 of the 9 partial products, the dropped ones < 2^-24 of the "
                           "product) on the bf16 MFMA pipe -- see verified.gram for the measured accuracy")
                          if int(lib.vivit_gemm_split_mode()) else "fp32 MFMA throughout",
            "data": "synthetic (seeded random-init MLP, uniform random inputs, materialised exact sqrt-GGN factors)",
            "config": {
                "workload": args.workload,
                "n": n, "P": P_total, "eigenvectors": vectors, "pairs_per_step": n,
                "level": "kernel launchers (vivit_amd.kernels) on resident factors; hook scheduling, criterion sync and "
                         "back-projection of the public API are outside the metric",
                "parallelism": (f"data parallel x{world}: batch-sharded factors -> all-to-all to parameter shards -> "
                                f"partial SYRK -> RCCL all-reduce; symeig: replicated reduction/D&C, back-transformations "
                                f"sharded x{world} + all-gather") if world > 1 else "single GPU",
            },
            "phases": {"exchange_s": exch_s, "gram_s": gram_s, "allreduce_s": ar_s, "symeig_s": eig_s},
            "roofline": roofline,
            "roofline_phases": phases_roof,
            "kernels": secondary,
        }
        if verified is not None:
            out["verified"] = verified
        if secondary_lines is not None:
            out["secondary"] = secondary_lines
        if world == 1 and not args.no_configs:
            # BASELINE configs 1, 3, 4, 5 (one GPU) through the public API on real factors (bench_configs.py)
            import bench_configs

            del facs[:]
            G = w = Z = None
            torch.cuda.empty_cache()
            out["configs"] = bench_configs.run_configs(device, progress=_progress)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dims, C, n, P_total,
                                               batches=tuple(sorted({min(int(b), batch) for b in args.cpu_batches.split(",")})),
                                               eig_batches=tuple(sorted({min(int(b), batch) for b in args.cpu_eig_batches.split(",")})))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
