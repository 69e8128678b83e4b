"""Pins the arithmetic of the default Gram path (fp32 products formed from the exact three-way bf16 split of the
operands, 6 of 9 partial products on the bf16 MFMA pipe): it must be fp32-class, i.e. indistinguishable from the fp32
MFMA kernel (a k-ordered fp32 fma chain, the successor of the reference's sgemm behind ``einsum``,
vivit/utils/gram.py:230-232) in the only unit where the difference between "6 partial products" and "3 partial products"
is visible -- the error per term of a random-walk sum (see gram_precision_child.py).  The knob VIVIT_GEMM_SPLIT is read
once per process, so every setting runs in a child process; ``=3`` (per-product error 2^-16) MUST fail the same
criterion, which proves the criterion has teeth.

Second part: the input-range contract of include/vivit_hip.h (inf / NaN / values beyond the bf16 range / tiny values
take the fp32 MFMA kernel for their column chunk and therefore behave exactly like IEEE fp32).
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

HERE = os.path.dirname(os.path.abspath(__file__))
# (n, K): 5120 rows = 210 lower 256-tiles -> chunked 256-tile launch; 1280 rows = 15 tiles -> split-K launch
TILE_CASES = ["5120:2064", "5120:65552", "5120:401408", "5120:2064:mixed"]
SPLITK_CASES = ["1280:65552", "1280:401408", "1280:65552:mixed"]
NINE_CASES = ["5120:2064", "5120:65552", "5120:2064:mixed"]


def _child(mode, cases, tmp_path):
    out = tmp_path / f"prec_{mode}.json"
    env = dict(os.environ, VIVIT_GEMM_SPLIT=str(mode))
    subprocess.run([sys.executable, os.path.join(HERE, "gram_precision_child.py"), str(out)] + cases, env=env, check=True,
                   timeout=900)
    res = json.loads(out.read_text())
    assert res["split_mode"] == mode
    return {(c["n"], c["K"], c["mixed"]): c for c in res["cases"]}


@pytest.fixture(scope="module")
def stats(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("gram_precision")
    return {
        0: _child(0, TILE_CASES + SPLITK_CASES, tmp),   # fp32 MFMA kernels: the yardstick
        6: _child(6, TILE_CASES + SPLITK_CASES, tmp),   # default
        3: _child(3, TILE_CASES, tmp),                  # three partial products: must be caught
        9: _child(9, NINE_CASES, tmp),                  # all nine partial products (32x32x16 MFMAs; optional mode)
    }


def _key(case):
    p = case.split(":")
    return int(p[0]), int(p[1]), len(p) > 2


@pytest.mark.parametrize("case", TILE_CASES + SPLITK_CASES)
def test_default_path_is_fp32_class(stats, case):
    f32, bx = stats[0][_key(case)], stats[6][_key(case)]
    print(case, "fp32 MFMA:", f32, "\nbf16 x 6:", bx)
    assert bx["symmetric"] and bx["finite"] and bx["entries"] >= 500
    # random-walk units; 2x the fp32 MFMA kernel's own rounding on the same data
    for k in ("offdiag_rms", "mirror_rms", "diag_rms"):
        assert bx[k] <= 2.0 * f32[k], (k, bx[k], f32[k])
    for k in ("offdiag_max", "mirror_max", "diag_max"):
        assert bx[k] <= 2.5 * f32[k], (k, bx[k], f32[k])
    # and in absolute terms: fp32 accumulation of a (two-level) random-walk sum stays below 2.5e-6 per term, the mean
    # relative error of the sums of squares on the diagonal (where truncation would show as a bias) below 2e-7
    assert bx["offdiag_rms"] <= 2.5e-6
    assert abs(bx["diag_mean"]) <= 2e-7 + 2.0 * abs(f32["diag_mean"])


@pytest.mark.parametrize("case", NINE_CASES)
def test_nine_partial_products_mode(stats, case):
    """VIVIT_GEMM_SPLIT=9 (the three dropped partial products added; the kernel instance that stayed on 32x32x16 MFMAs when the default
    moved to the fused 16x16x32 form): fp32-class by the same criterion and no worse than the default."""
    f32, b9, b6 = stats[0][_key(case)], stats[9][_key(case)], stats[6][_key(case)]
    print(case, "fp32 MFMA:", f32["offdiag_rms"], "bf16 x 9:", b9["offdiag_rms"], "bf16 x 6:", b6["offdiag_rms"])
    assert b9["symmetric"] and b9["finite"] and b9["entries"] >= 500
    for k in ("offdiag_rms", "mirror_rms", "diag_rms"):
        assert b9[k] <= 2.0 * f32[k], (k, b9[k], f32[k])
    assert b9["offdiag_rms"] <= 1.1 * b6["offdiag_rms"]


@pytest.mark.parametrize("case", TILE_CASES)
def test_three_partial_products_would_be_caught(stats, case):
    """The same criterion applied to VIVIT_GEMM_SPLIT=3 (hi hi + hi mid + mid hi only: every product off by 2^-16) FAILS:
    the dropped partial products alone are ~4e-6 per term, 2.7 times the whole error of the fp32 MFMA kernel."""
    f32, b3, b6 = stats[0][_key(case)], stats[3][_key(case)], stats[6][_key(case)]
    print(case, "fp32 MFMA:", f32["offdiag_rms"], "bf16 x 3:", b3["offdiag_rms"], "bf16 x 6:", b6["offdiag_rms"])
    assert b3["offdiag_rms"] > 2.0 * f32["offdiag_rms"]      # the criterion of test_default_path_is_fp32_class, violated
    assert b3["offdiag_rms"] > 1.8 * b6["offdiag_rms"]      # (1.96 at K = 401 408 since the diagonal TILES run the 4096-k chains of all others)
    assert b3["offdiag_rms"] > 3e-6


@pytest.mark.parametrize("n,K", [(5120, 2064), (5120, 2 * 65536 + 2064), (1280, 65552)],
                         ids=["tile-1chunk", "tile-3chunks", "splitk"])
def test_input_range_contract(n, K):
    """inf / NaN / 3.4e38 / 1e-39 inputs through ``gram_syrk`` on the bf16-pipe shapes: IEEE fp32 results."""
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    A = torch.randn((n, K), generator=g, device=dev)
    kc = K - 1000  # special columns sit in the LAST chunk; the others must stay clean
    A[:, kc + 7] = torch.rand(n, generator=g, device=dev) * 1.8 - 0.9
    A[7, kc + 100] = float("inf")
    A[300, kc + 5] = float("nan")
    A[600] = 0.0
    A[600, kc + 7] = 3.4e38          # above the bf16 range (bf16(3.4e38) = inf), finite in fp32
    A[900] = 0.0
    A[900, kc + 9] = 1e-39           # fp32 subnormal
    A[1000, kc + 9] = 1e30
    G = kernels.gram_syrk(A)
    torch.cuda.synchronize()
    assert torch.equal(G.isnan(), G.T.isnan()) and torch.equal(G.nan_to_num(0.0, 1.0, -1.0), G.T.nan_to_num(0.0, 1.0, -1.0))
    special = [7, 300, 600, 900, 1000]
    clean = torch.ones(n, dtype=torch.bool, device=dev)
    clean[special] = False
    # rows without special values: finite everywhere and as accurate as ever
    Gc = G[clean][:, clean]
    assert torch.isfinite(Gc).all()
    ri = torch.arange(0, n, 37, device=dev)
    ri = ri[clean[ri]]
    ref = A[ri].double() @ A[ri].double().T
    d = ref.diagonal().sqrt()
    err = ((G[ri][:, ri].double() - ref).abs() / (d[:, None] * d[None, :])).max().item()
    assert err <= 5e-6, err   # relative to sqrt(G_ii G_jj), the scale of an entry's terms
    # inf row: +-inf with the sign of the partner's entry (inf * x + finite), NaN against the NaN row and against rows
    # that hold an exact zero in that column (inf * 0)
    row = G[7]
    partner = A[:, kc + 100]
    ok = clean.clone()
    expect = torch.where(partner[ok] > 0, float("inf"), -float("inf"))
    assert torch.equal(row[ok], expect.float())
    assert row[7] == float("inf") and row[300].isnan() and row[600].isnan() and row[900].isnan()
    # NaN row: NaN everywhere
    assert G[300].isnan().all() and G[:, 300].isnan().all()
    # 3.4e38 * x with |x| < 0.9 is finite in fp32 (a bf16 'hi' piece of inf would have made it NaN)
    x = A[:, kc + 7]
    got, want = G[600][ok].double(), 3.4e38 * x[ok].double()
    assert torch.isfinite(got).all()
    assert ((got - want).abs() <= 1e-6 * want.abs()).all()
    assert G[600, 600] == float("inf")
    # fp32 subnormal input: multiplied as a subnormal (1e-39 * 1e30 = 1e-9), not flushed, no lost pieces
    tiny = A[900, kc + 9].double().item()   # the subnormal fp32 value actually stored
    want = tiny * A[1000, kc + 9].double().item()
    assert abs(G[900, 1000].item() - want) <= 2e-7 * want, (G[900, 1000].item(), want)
    sub = G[900][ok].double()
    assert ((sub - tiny * A[:, kc + 9][ok].double()).abs() <= 2e-45).all()
