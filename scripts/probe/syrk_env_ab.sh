#!/bin/bash
# Same-box timing of the headline-shaped Gram SYRK with the product library under different environment settings, interleaved.
#   scripts/probe/syrk_env_ab.sh "VIVIT_BX_SYNC=1" "VIVIT_BX_FLUSH=8192" ...      (the empty setting runs first in every round)
cd "$(dirname "$0")/../.."
run() { env $1 python - "$1" <<'PY' 2>&1 | grep -v "amdgpu.ids\|arn"
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
print(f"[{sys.argv[1] or 'default'}]", " | ".join(res), flush=True)
PY
}
for rep in 1 2; do
  run ""
  for s in "$@"; do run "$s"; done
done
