import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_q2_slide_gpu import chase, apply_reference, DEV
from vivit_amd import kernels
for n, nrows in ((196, 16), (516, 130), (1024, 300)):
    band, d, e, R2, tau2 = chase(n, n)
    rng = np.random.default_rng(n + 1)
    Z0 = (rng.standard_normal((nrows, n)) / np.sqrt(n)).astype(np.float32)
    ref = apply_reference(Z0, R2.cpu().numpy(), tau2.cpu().numpy())
    print("launching", n, nrows, flush=True)
    got = kernels.q2_apply_(torch.from_numpy(Z0).to(DEV).clone(), R2, tau2, mode=1)
    torch.cuda.synchronize()
    got = got.cpu().double().numpy()
    print(n, nrows, "err", np.abs(got - ref).max() / np.abs(ref).max(), flush=True)
