"""Per-CU timeline of the bf16-pipe Gram SYRK (diagnostic build scripts/probe/libstamp2.so: scripts/probe/variants.sh gemm_f32 stamp2
-DBX_STAMP=2): every workgroup of the last non-mirroring chunk launch stamps s_memrealtime (100 MHz) at entry, at the start and the
end of its K loop and behind its flush, together with the CU it ran on.  Prints the medians of prologue / K loop / flush and of the
gap between consecutive workgroups of a CU.   usage: python bx_timeline.py [randn|half]"""
import ctypes, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ["VIVIT_HIP_ALLOW_STALE"] = "1"
import torch
import vivit_amd._lib as L
L.LIB_PATH = os.path.abspath(os.environ.get("VIVIT_LIB", os.path.join(ROOT, "scripts", "probe", "libstamp2.so")))
from vivit_amd import kernels
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "half"
n, p = 40960, 32768
A = torch.randn(n, p, device=dev)
if kind == "half":
    A.mul_((torch.rand(n, p // 784 + 1, device=dev) < 0.5).repeat_interleave(784, 1)[:, :p])
cap = 1 << 14
buf = torch.zeros(8 * cap, dtype=torch.int64, device=dev)
lib = L.load()
lib.vivit_debug_bx_stamp_buffer.restype = ctypes.c_int
lib.vivit_debug_bx_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_uint]
assert lib.vivit_debug_bx_stamp_buffer(buf.data_ptr(), cap) == 0
G = torch.zeros(n, n, device=dev)
for _ in range(3):
    kernels.gram_syrk(A, out=G)
torch.cuda.synchronize()
st = buf.view(cap, 8).cpu()
ok = st[:, 5] > 0
st = st[ok]
us = lambda x: x.double() / 100.0
entry, loop0, loop1, exit_ = st[:, 2], st[:, 3], st[:, 4], st[:, 5]
hw = st[:, 6]
cu = ((hw >> 32) & 15) * 4096 + ((hw >> 8) & 0xF) + (((hw >> 12) & 1) << 4) + (((hw >> 13) & 7) << 5)   # (xcc, se, sh, cu)
q = lambda x: f"median {x.median().item():.1f} (p10 {x.quantile(0.1).item():.1f}, p90 {x.quantile(0.9).item():.1f})"
print(f"{kind}: {int(ok.sum())} workgroups on {len(torch.unique(cu))} CUs; launch span {us(exit_.max() - entry.min()).item():.0f} us")
print("  prologue (entry -> first tile)  us:", q(us(loop0 - entry)))
print("  K loop                          us:", q(us(loop1 - loop0)))
ghz = st[:, 0].double() / (st[:, 1].double() * 10.0)      # s_memtime (core clock) over s_memrealtime (100 MHz) across the K loop
print("  clock inside the K loop        GHz:", q(ghz), "| core cycles per K tile (256 per 4096-column launch):", q(st[:, 0].double() / 256))
print("  flush (loop end -> in memory)   us:", q(us(exit_ - loop1)), "| of which until wave 0 has issued its last store:", q(us(st[:, 7] - loop1)))
gaps = []
for c in torch.unique(cu):
    m = cu == c
    e, x = entry[m], exit_[m]
    o = torch.argsort(e)
    e, x = e[o], x[o]
    gaps.append(us(e[1:] - x[:-1]))
gaps = torch.cat(gaps)
print("  gap exit -> next entry on a CU  us:", q(gaps))
t0 = entry.min()
first = us(torch.stack([entry[cu == c].min() for c in torch.unique(cu)]) - t0)
last = us(exit_.max() - torch.stack([exit_[cu == c].max() for c in torch.unique(cu)]))
print("  first entry of a CU after launch start us:", q(first), "| idle tail of a CU us:", q(last))
