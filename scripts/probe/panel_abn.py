"""Interleaved timing of the 64-row streaming product for several library builds on one box (each in its own child process,
two rounds; white-noise and smooth operands).   python panel_abn.py lib1.so lib2.so ..."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
child = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "..", ".."))
from vivit_amd import kernels
m = 40960; real = len(sys.argv) > 1
A = torch.randn(64, m, device="cuda"); B = torch.randn(m, m, device="cuda")
if real: B = torch.cumsum(B, 1) / m ** 0.5
out = torch.empty(64, m, device="cuda")
for _ in range(3): kernels.gemm_nn(A, B, out=out)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): kernels.gemm_nn(A, B, out=out)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 20:.3f}")
''' % here
for rnd in range(2):
    for lib in sys.argv[1:]:
        res = []
        for extra in ([], ["smooth"]):
            env = dict(os.environ, VIVIT_HIP_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, "-c", child] + extra, env=env, capture_output=True, text=True)
            res.append(out.stdout.strip() or out.stderr[-200:])
        print(rnd, os.path.basename(lib), "randn", res[0], "ms | smooth", res[1], "ms", flush=True)
