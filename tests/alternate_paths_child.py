"""Child process of tests/test_alternate_paths_gpu.py: the kernels that have two implementations behind one entry point, run
with the environment the parent set (VIVIT_GEMM64_BX / VIVIT_CONV_MFMA are read once per process); results as a .pt file.

usage: python alternate_paths_child.py OUT.pt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vivit_amd import kernels  # noqa: E402

DEV = torch.device("cuda:0")


def run():
    g = torch.Generator(device=DEV).manual_seed(5)
    out = {}
    # 64-row streaming product (band reduction's panel product): bf16 pipe with exact splits | fp32 MFMA
    A = torch.randn(64, 4112, generator=g, device=DEV)
    B = torch.randn(4112, 2304, generator=g, device=DEV)
    out["panel_nn"] = kernels.gemm_nn(A, B)
    out["panel_nt"] = kernels.gemm_nt(A, B.T.contiguous())
    # convolution rules on ResNet-32's three stages and a strided layer: matrix pipe | scalar kernels
    for name, (cin, cout, hw, s) in {"s1": (16, 16, 32, 1), "s2": (32, 32, 16, 1), "s3": (64, 64, 8, 1), "down": (16, 32, 32, 2),
                                     "wide": (24, 160, 8, 1)}.items():
        x = torch.randn(5, cin, hw, hw, generator=g, device=DEV)
        w = torch.randn(cout, cin, 3, 3, generator=g, device=DEV)
        oh = (hw + 2 - 3) // s + 1
        M = torch.randn(2, 5, cout, oh, oh, generator=g, device=DEV)
        out["w_" + name] = kernels.conv2d_weight_mjp(M, x, (3, 3), (s, s), (1, 1), (1, 1))
        try:
            out["j_" + name] = kernels.conv2d_jac_t(M, w, (hw, hw), (s, s), (1, 1), (1, 1))
        except Exception as exc:  # the scalar kernel refuses 160 output channels: the parent expects exactly that
            out["j_" + name] = str(getattr(exc, "status", exc))
    return {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in out.items()}


if __name__ == "__main__":
    torch.save(run(), sys.argv[1])
