"""Child process of tests/test_determinism_gpu.py: the hot path's products and eigensolves on seeded inputs; SHA-1 of
every output as JSON.  The parent runs the same cases itself (twice) and compares: bit-reproducibility from call to call
AND from process to process (fresh allocations at other addresses, fresh workgroup-to-CU placement).

usage: python determinism_child.py OUT.json [case ...]
"""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vivit_amd import kernels  # noqa: E402

DEV = torch.device("cuda:0")


def sha(*tensors):
    h = hashlib.sha1()
    for t in tensors:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def factor(n, K, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(n, K, generator=g, device=DEV) * (1.0 / 64.0)


def sym(n, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    V = torch.randn(n, n + 64, generator=g, device=DEV) / float(n) ** 0.5
    return kernels.gram_syrk(V)


def case_syrk256_k65552():
    # 256-tile bf16-pipe launch (210 lower tiles), ragged K: 16 chains of 4096 k flushed by atomics + tail
    return sha(kernels.gram_syrk(factor(5120, 65552, 1)))


def case_syrk256_k401408():
    # the headline contraction length: 98 launches of one 4096-column chain each, every one added into C by atomics
    return sha(kernels.gram_syrk(factor(5120, 401408, 2)))


def case_syrk_accumulate():
    # gram += gram_p (beta = 1): the last flush of a tile reads C
    A = factor(5120, 20000, 3)
    G = kernels.gram_syrk(A[:, :12000].contiguous())
    kernels.gram_syrk(A[:, 12000:].contiguous(), out=G, beta=1.0)
    return sha(G)


def case_gemm_nt_splitk():
    # small output, deep contraction: split-K slab + fixed-order reduce
    A, B = factor(1024, 401408, 4), factor(40, 401408, 5)
    return sha(kernels.gemm_nt(A, B), kernels.gram_syrk(A))


def case_symeig_two_stage_4100():
    w, Z = kernels.symeig(sym(4100, 6), eigenvectors=True)
    return sha(w, Z)


def case_symeig_two_stage_8192():
    w, Z = kernels.symeig(sym(8192, 7), eigenvectors=True)
    return sha(w, Z)


def case_symeig_values_8192():
    return sha(kernels.symeig(sym(8192, 7), eigenvectors=False)[0])


def case_symeig_reduce_select():
    plan = kernels.symeig_reduce(sym(4100, 8))
    idx = torch.tensor([0, 17, 4000, 4090, 4099], device=DEV, dtype=torch.int32)
    Zt = plan.select(idx)
    return sha(plan.evals, Zt)


def case_panel_product_bx():
    # the band reduction's 64-row streaming product on the bf16 pipe: split-K slab (fixed-order reduce), an odd number of K
    # tiles per split, chains closed in registers
    A, B = factor(64, 6160, 8), factor(6160, 2560, 9)
    return sha(kernels.gemm_nn(A, B))


def case_conv_rules():
    # convolution weight rule on the matrix pipe (split position range with the fixed-order LDS reduction: 16 -> 16 @ 32 x 32)
    # and the input rules (scalar kernel; the matrix-pipe kernel with a split contraction for 160 output channels)
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(6, 16, 32, 32, generator=g, device=DEV)
    M = torch.randn(2, 6, 16, 32, 32, generator=g, device=DEV)
    w = torch.randn(16, 16, 3, 3, generator=g, device=DEV)
    a = kernels.conv2d_weight_mjp(M, x, (3, 3), (1, 1), (1, 1), (1, 1))
    b = kernels.conv2d_jac_t(M, w, (32, 32), (1, 1), (1, 1), (1, 1))
    M2 = torch.randn(2, 6, 160, 8, 8, generator=g, device=DEV)
    w2 = torch.randn(160, 24, 3, 3, generator=g, device=DEV)
    c = kernels.conv2d_jac_t(M2, w2, (8, 8), (1, 1), (1, 1), (1, 1))
    return sha(a, b, c)


CASES = {k[5:]: v for k, v in globals().items() if k.startswith("case_")}


def run(names):
    out = {}
    for name in names:
        out[name] = CASES[name]()
        kernels._WORKSPACES.clear()
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    # perturb the allocator so that the operands sit at other addresses than in the parent
    pad = torch.empty(123_457_024, dtype=torch.uint8, device=DEV)
    names = sys.argv[2:] or sorted(CASES)
    res = run(names)
    del pad
    with open(sys.argv[1], "w") as f:
        json.dump(res, f)
