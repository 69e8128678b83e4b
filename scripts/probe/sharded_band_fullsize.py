"""The sharded band reduction (vivit_amd/distributed.py:sy2sb_sharded_) at a FULL size on one GPU: a forced one-rank world on backend
nccl (VIVIT_DIST_FORCE_COLLECTIVES=1: every collective is issued through RCCL) runs the same panel loop an 8-rank job runs -- all
rows local -- so that its Python / launch overhead per panel and its results can be measured where only one GPU exists.
usage: python scripts/probe/sharded_band_fullsize.py [n=20480]"""
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29731")
os.environ["VIVIT_DIST_FORCE_COLLECTIVES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import bench
from vivit_amd import distributed as vd, kernels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator(device=dev).manual_seed(0)
V = torch.randn(n, 4096, device=dev, generator=g) * torch.logspace(0, -3, 4096, device=dev)   # a decaying GGN-like spectrum, rank 4096
G = kernels.gram_syrk(V)
del V
for tag, kw in (("replicated reduction (single-GPU kernels)", {"sharded_reduction": False}), ("sharded reduction, one forced rank", {"sharded_reduction": True})):
    vd.LAST_SHARDED_COLLECTIVES.clear()
    w, Z = vd.symeig(G.clone(), overwrite=True, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w, Z = vd.symeig(G.clone(), overwrite=True, **kw)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    ve = bench.verify_symeig(G, w, Z)
    print(f"n={n} {tag}: {t:.3f} s; residual_2norm_fp64 {ve['residual_2norm_fp64']:.2e}, orth {ve['orth_err']:.2e}, trace {ve['trace_err']:.1e}; "
          f"collectives {dict(vd.LAST_SHARDED_COLLECTIVES)}", flush=True)
dist.destroy_process_group()
