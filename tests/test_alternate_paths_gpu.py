"""Two implementations behind one entry point must agree: the band reduction's 64-row streaming product (bf16 pipe with
exact operand splits, default | fp32 MFMA, VIVIT_GEMM64_BX=0) and the convolution rules of the factor provider (fp32 matrix
pipe, default | scalar kernels, VIVIT_CONV_MFMA=0).  The switches are read once per process, so the alternative runs in a
child; both are also compared with fp64 references."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def both(tmp_path_factory):
    sys.path.insert(0, HERE)
    import alternate_paths_child as child

    out = str(tmp_path_factory.mktemp("alt") / "alt.pt")
    env = dict(os.environ, VIVIT_GEMM64_BX="0", VIVIT_CONV_MFMA="0")
    proc = subprocess.run([sys.executable, os.path.join(HERE, "alternate_paths_child.py"), out], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stdout[-3000:]
    return child.run(), torch.load(out)


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()


@pytest.mark.parametrize("key", ["panel_nn", "panel_nt", "w_s1", "w_s2", "w_s3", "w_down", "w_wide", "j_s1", "j_s2", "j_s3", "j_down"])
def test_default_and_alternative_agree(key, both):
    default, alt = both
    assert torch.is_tensor(default[key]) and torch.is_tensor(alt[key])
    assert default[key].shape == alt[key].shape
    # fp32 sums of 4112 (panel) or <= 1024 (rules) products in two different orders
    assert _rel(default[key], alt[key]) < 2e-5


def test_wide_filter_slice_only_runs_on_the_matrix_pipe(both):
    """160 output channels x 3 x 3 = 1440 > 1024: the scalar input rule refuses (VIVIT_E_UNSUPPORTED = -4), the default build
    serves it with the matrix-pipe kernel -- against an fp64 transposed convolution."""
    default, alt = both
    assert alt["j_wide"] == "-4"
    g = torch.Generator(device="cuda:0").manual_seed(5)
    # regenerate the child's inputs of that case: same generator sequence as alternate_paths_child.run
    A = torch.randn(64, 4112, generator=g, device="cuda:0"); B = torch.randn(4112, 2304, generator=g, device="cuda:0")
    del A, B
    ref = None
    for cin, cout, hw, s in [(16, 16, 32, 1), (32, 32, 16, 1), (64, 64, 8, 1), (16, 32, 32, 2), (24, 160, 8, 1)]:
        x = torch.randn(5, cin, hw, hw, generator=g, device="cuda:0")
        w = torch.randn(cout, cin, 3, 3, generator=g, device="cuda:0")
        oh = (hw + 2 - 3) // s + 1
        M = torch.randn(2, 5, cout, oh, oh, generator=g, device="cuda:0")
        if cout == 160:
            ref = torch.stack([torch.nn.functional.conv_transpose2d(M[v].double(), w.double(), stride=s, padding=1) for v in range(2)])
    assert _rel(default["j_wide"].cuda(), ref) < 1e-5
