"""A/B timing of the 64-row streaming product between library builds on one box, interleaved: each library in its own
child process (the library is chosen at import), three rounds.   python panel_ab.py libA.so libB.so [m]"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
m = sys.argv[3] if len(sys.argv) > 3 else "40960"
child = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "..", ".."))
from vivit_amd import kernels
m = int(sys.argv[1]); real = len(sys.argv) > 2
A = torch.randn(64, m, device="cuda"); B = torch.randn(m, m, device="cuda")
if real:  # a smooth matrix instead of white noise (the chip's clock depends on the data's toggle rate)
    B = torch.cumsum(B, 1) / m ** 0.5
out = torch.empty(64, m, device="cuda")
for _ in range(3): kernels.gemm_nn(A, B, out=out)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): kernels.gemm_nn(A, B, out=out)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 20:.3f}")
''' % here
for rnd in range(3):
    for lib in sys.argv[1:3]:
        for extra in ([], ["smooth"]):
            env = dict(os.environ, VIVIT_HIP_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, "-c", child, m] + extra, env=env, capture_output=True, text=True)
            print(rnd, os.path.basename(lib), "smooth" if extra else "randn ", out.stdout.strip() or out.stderr[-300:], "ms", flush=True)
