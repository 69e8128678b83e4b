"""Child process of tests/test_bx_asm_gpu.py: products on the 256-tile bf16-pipe kernel with the K loop the parent chose
(VIVIT_BX_ASM is read once per process); results as a .pt file.

usage: python bx_asm_child.py OUT.pt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vivit_amd import kernels  # noqa: E402

DEV = torch.device("cuda:0")


def run():
    g = torch.Generator(device=DEV).manual_seed(11)
    out = {}
    # Gram SYRKs: several 4096-column chains per product (one launch each), a ragged last chunk, edge tiles, mirrored tiles
    for name, (n, p) in {"syrk_a": (1536, 40000), "syrk_b": (3000, 9000), "syrk_c": (4096, 16400), "syrk_d": (777, 70000)}.items():
        out[name] = kernels.gram_syrk(torch.randn(n, p, generator=g, device=DEV))
    # NT products with beta (Q1's Zt -= W Y^T, the trailing updates' shape): one chain, the final flush reads C
    for name, (m, n, k) in {"nt_a": (2048, 4096, 2048), "nt_b": (1000, 3000, 5000), "nt_c": (4096, 4096, 512), "nt_d": (2560, 2560, 8192)}.items():
        A = torch.randn(m, k, generator=g, device=DEV)
        B = torch.randn(n, k, generator=g, device=DEV)
        C = torch.randn(m, n, generator=g, device=DEV)
        out[name] = kernels.gemm_nt(A, B, out=C, alpha=-1.0, beta=1.0)
    # the eigensolver's own products ride the same kernel: a two-stage solve end to end
    A = torch.randn(3072, 3500, generator=g, device=DEV)
    w, Z = kernels.symeig(kernels.gram_syrk(A), eigenvectors=True)
    out["eig_w"], out["eig_Z"] = w, Z
    return {k: v.cpu() for k, v in out.items()}


if __name__ == "__main__":
    torch.save(run(), sys.argv[1])
