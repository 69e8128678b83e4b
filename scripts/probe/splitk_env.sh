#!/bin/bash
# The config-1-shaped Gram SYRK (n = 1280, K = 401 408: split-K form of the bf16-pipe tile kernel) under environment settings.
cd "$(dirname "$0")/../.."
run() { env $1 python - "$1" <<'PY' 2>&1 | grep -v "amdgpu.ids\|arn"
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
out = []
for n, K in ((1280, 401408), (2048, 40960)):
    A = torch.randn(n, K, device=dev)
    G = kernels.gram_syrk(A); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); kernels.gram_syrk(A, out=G); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[3]
    out.append(f"n={n} K={K}: {t*1e3:.3f} ms = {n*(n+1)*K/t/1e12:.1f} TF")
print(f"[{sys.argv[1] or 'default'}]", " | ".join(out), flush=True)
PY
}
run ""
for s in "$@"; do run "$s"; done
