"""The HEADLINE shape itself (BASELINE config 2: n = 40 960, P = 407 050, the factors bench.py times): Gram
matrix against fp64 dot products of sampled entries (all tile classes of the 256-tile SYRK incl. the 49
read-modify-write flushes of C over K = 401 408), exact symmetry, trace; then the full eigendecomposition's
properties on THAT matrix.  The oracle cannot run at this size; torch fp64 ops on the device are the checker."""
import pytest
import torch

import bench

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]


@pytest.mark.parametrize("workload", ["mlp784-512-10_b1024", "mlp784-512-10_b4096"])
def test_headline_gram_and_symeig(workload):
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    dims, batch, C = bench.WORKLOADS[workload]
    n = C * batch
    P = dims[0] * dims[1] + dims[1] + dims[1] * dims[2] + dims[2]
    free, _ = torch.cuda.mem_get_info()
    if free < 4 * n * P + 10 * 4 * n * n + (8 << 30):
        pytest.skip("not enough free HBM for the headline-shape test")
    facs = bench.mlp_sqrt_ggn_factors(dims, batch, dev)
    assert sum(f.shape[1] for f in facs) == P and all(f.shape[0] == n for f in facs)
    G = torch.empty((n, n), dtype=torch.float32, device=dev)
    for k, A in enumerate(facs):
        kernels.gram_syrk(A, out=G, alpha=1.0, beta=0.0 if k == 0 else 1.0)
    vg = bench.verify_gram(facs, G, num=128)
    print(workload, "gram:", vg)
    assert vg["entries"] >= 10000
    assert vg["symmetric"]
    assert vg["entry_err"] <= bench.VERIFY_BOUNDS["entry_err"], vg
    assert vg["diag_err"] <= bench.VERIFY_BOUNDS["diag_err"], vg
    assert vg["trace_err"] <= bench.VERIFY_BOUNDS["trace_err"], vg
    del facs
    torch.cuda.empty_cache()
    w_only, _ = kernels.symeig(G, eigenvectors=False)
    w, Z = kernels.symeig(G, eigenvectors=True)
    ve = bench.verify_symeig(G, w, Z)
    print(workload, "symeig:", ve)
    assert ve["ascending"]
    assert (w - w_only).abs().max().item() <= 1e-5 * ve["lambda_max"]   # D&C against Sturm multisection
    assert ve["trace_err"] <= bench.VERIFY_BOUNDS["eig_trace_err"], ve
    assert ve["fro_err"] <= bench.VERIFY_BOUNDS["fro_err"], ve
    assert ve["orth_err"] <= bench.VERIFY_BOUNDS["orth_err"], ve                       # all of Z^T Z - I, fp64 accumulation
    assert ve["residual_err"] <= bench.VERIFY_BOUNDS["residual_err"], ve
    # BASELINE's tolerance, literally: max_i ||G z_i - w_i z_i||_2 <= 1e-5 lambda_max in fp64 => every eigenvalue has an exact
    # eigenvalue of G within 1e-5 lambda_max (the scope of test/linalg/test_eigvalsh.py:55-60)
    assert ve["residual_2norm_fp64"] <= bench.VERIFY_BOUNDS["residual_2norm_fp64"] == 1e-5, ve
    # the GGN of a C-class cross-entropy model has rank <= N (C - 1): the bottom N eigenvalues are rounding noise
    assert w[:batch].abs().max().item() <= 1e-5 * ve["lambda_max"]
