import glob, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vivit_amd import kernels
for f in sorted(glob.glob(os.path.join(ROOT, "scripts/probe/data/*.pt"))):
    H = torch.load(f).cuda()
    ref = torch.linalg.eigvalsh(H.double().cpu())
    try:
        w, Z = kernels.symeig(H, eigenvectors=True)
        res = (H @ Z - Z * w).abs().max().item() / ref[-1].item()
        print(os.path.basename(f), "ok err", ((w.cpu().double() - ref).abs().max() / ref[-1]).item(), "res", res, flush=True)
    except RuntimeError as e:
        print(os.path.basename(f), "FAILED", e, flush=True)
