import sys, time
sys.path.insert(0, "/root/repo")
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
for n in [8192]:
    M = torch.randn(n, n, device=dev); S = (M + M.T) / 2
    S[100, 200] = float("nan"); S[200, 100] = float("nan")
    for vec in [False, True]:
        t0 = time.time()
        try:
            w, Z = kernels.symeig(S, eigenvectors=vec)
            torch.cuda.synchronize()
            print(n, vec, "returned; nan in w:", bool(torch.isnan(w).any()), f"{time.time()-t0:.2f}s")
        except RuntimeError as e:
            print(n, vec, "RuntimeError:", str(e)[:80], f"{time.time()-t0:.2f}s")
    S2 = torch.zeros(n, n, device=dev)
    w, Z = kernels.symeig(S2, eigenvectors=True); torch.cuda.synchronize()
    print("zero matrix ok", float(w.abs().max()), float((Z.T @ Z - torch.eye(n, device=dev)).abs().max()))
    S3 = torch.eye(n, device=dev) * 3
    w, Z = kernels.symeig(S3, eigenvectors=True); torch.cuda.synchronize()
    print("identity ok", float((w - 3).abs().max()))
