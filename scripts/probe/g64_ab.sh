#!/bin/bash
# same-box A/B of the streaming panel product: product library (scalar-base requests) against -DG64_VADDR
cd /root/repo
run() { VIVIT_HIP_ALLOW_STALE=1 VIVIT_HIP_LIB=$1 python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|arn"
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from vivit_amd import kernels
res = []
for m in (40960, 20480, 8192):
    A = torch.randn(64, m, device="cuda"); B = torch.randn(m, m, device="cuda"); out = torch.empty(64, m, device="cuda")
    kernels.gemm_nn(A, B, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); kernels.gemm_nn(A, B, out=out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    res.append(f"m={m}: {t*1e6:.0f} us = {4.0*m*m/t/1e12:.2f} TB/s")
    ref = (A.double() @ B.double())
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    res.append(f"err {err:.1e}")
print(os.environ.get("VIVIT_HIP_LIB"), " | ".join(res), flush=True)
PY
}
for rep in 1 2; do run vivit_amd/libvivit_hip.so; run scripts/probe/libg64vaddr.so; done
