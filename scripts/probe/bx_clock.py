"""In-kernel clock of the bf16-pipe Gram SYRK (diagnostic build scripts/probe/libstamp.so: -DBX_STAMP): back-to-back
launches on the headline shape for >= 2 s, then the stamps of the LAST launch: median over workgroups of
d(s_memtime) / d(s_memrealtime) x 100 MHz.   usage: python bx_clock.py [randn|real|zeros]"""
import ctypes, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
import vivit_amd._lib as L
L.LIB_PATH = os.path.abspath(os.environ.get("VIVIT_LIB", os.path.join(ROOT, "scripts", "probe", "libstamp.so")))
from vivit_amd import kernels
import bench
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "real"
n = 40960
if kind == "real":
    A = bench.mlp_sqrt_ggn_factors((784, 512, 10), 4096, dev)[2]        # the first-layer weight's factor, K = 401 408
elif kind == "zeros":
    A = torch.zeros(n, 401408, device=dev)
else:
    A = torch.randn(n, 401408, device=dev)
cap = 1 << 16
buf = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
lib = L.load()
lib.vivit_debug_bx_stamp_buffer.restype = ctypes.c_int
lib.vivit_debug_bx_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_uint]
assert lib.vivit_debug_bx_stamp_buffer(buf.data_ptr(), cap) == 0
G = torch.empty(n, n, device=dev)
t0 = time.perf_counter()
reps = 0
while time.perf_counter() - t0 < 6.0 or reps < 2:
    kernels.gram_syrk(A, out=G)
    torch.cuda.synchronize()
    reps += 1
dt = (time.perf_counter() - t0) / reps
st = buf.view(cap, 2)
ok = st[:, 1] > 0
clk = (st[ok, 0].double() / st[ok, 1].double() * 100e6 / 1e9)
wall = st[ok, 1].double() / 100e6 * 1e3
print(f"{kind}: {reps} SYRKs, {dt*1e3:.1f} ms each, {n*(n+1)*A.shape[1]/dt/1e12:.1f} TFLOP/s; stamps of the last chunk launch: "
      f"{int(ok.sum())} workgroups, in-kernel clock median {clk.median().item():.3f} GHz (p10 {clk.quantile(0.1).item():.3f}, "
      f"p90 {clk.quantile(0.9).item():.3f}); K-loop wall per tile median {wall.median().item():.2f} ms")
