#!/bin/bash
# Same-box timing of the headline-shaped Gram SYRK (n = 40 960, P = 131 072 of N(0,1) and of half-zero data) with the product
# library and with variant libraries scripts/probe/lib<tag>.so (scripts/probe/variants.sh gemm_f32 <tag> -D...), interleaved.
#   scripts/probe/syrk_ab.sh tag1 [tag2 ...]
cd "$(dirname "$0")/../.."
run() { VIVIT_HIP_ALLOW_STALE=1 VIVIT_HIP_LIB=$1 python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|arn"
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n, p = 40960, 131072
G = torch.empty(n, n, device=dev)
A = torch.randn(n, p, device=dev)
res = []
for kind in ("randn", "half zeros"):
    if kind == "half zeros":
        A.mul_((torch.rand(n, p // 784 + 1, device=dev) < 0.5).repeat_interleave(784, 1)[:, :p])
    kernels.gram_syrk(A, out=G); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); kernels.gram_syrk(A, out=G); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    res.append(f"{kind}: {t*1e3:.1f} ms = {n*(n+1)*p/t/1e12:.1f} TF")
# a rank-1024 update C -= V W^T of a 20480 x 20480 block (the band reduction's trailing update / Q1's shape: one 64-tile chain + flush)
del A, G
m, k = 20480, 1024
V = torch.randn(m, k, device=dev); W = torch.randn(m, k, device=dev); C = torch.randn(m, m, device=dev)
kernels.gemm_nt(V, W, out=C, alpha=-1.0, beta=1.0); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    kernels.gemm_nt(V, W, out=C, alpha=-1.0, beta=1.0)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 10
res.append(f"rank-1024 update of 20480^2: {t*1e3:.2f} ms = {2*m*m*k/t/1e12:.1f} TF")
print(os.environ.get("VIVIT_HIP_LIB"), " | ".join(res), flush=True)
PY
}
for rep in 1 2; do
  run vivit_amd/libvivit_hip.so
  for t in "$@"; do run scripts/probe/lib$t.so; done
done
