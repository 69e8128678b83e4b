"""Bulge chasing alone at n = 40960 (random band) against a (possibly experimental) build of the library:
python run_variant_sb2st.py <path/to/libvivit_hip.so>.  Variants built with -DSB2ST_VARIANT=1 (no compute) or
=2 (no block loads/stores) give invalid results; they attribute the 11.5 us wavefront step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vivit_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40960
AB = torch.randn(n, 129, device=dev)
AB[:, :64] = 0
kernels.sb2st(AB[:4096].contiguous()); torch.cuda.synchronize()
t0 = time.perf_counter(); kernels.sb2st(AB); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(os.path.basename(sys.argv[1]), f"{dt*1e3:.0f} ms  {dt / (2 * n) * 1e6:.2f} us per wavefront step", flush=True)
