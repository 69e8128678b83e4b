"""Would two chunk launches of the Gram SYRK running side by side (even / odd chunks accumulating into two matrices on two streams) fill the
end-of-launch tail and hide the split passes?  Two independent SYRKs (n = 40 960, P = 32 768 each, half-zero data) back to back on one
stream against the same two on two streams at once."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vivit_amd import kernels
dev = torch.device("cuda:0")
n, p = 40960, 32768
A = [torch.randn(n, p, device=dev) for _ in range(2)]
for a in A:
    a.mul_((torch.rand(n, p // 784 + 1, device=dev) < 0.5).repeat_interleave(784, 1)[:, :p])
G = [torch.empty(n, n, device=dev) for _ in range(2)]
S = [torch.cuda.Stream(), torch.cuda.Stream()]
def seq():
    with torch.cuda.stream(S[0]):
        kernels.gram_syrk(A[0], out=G[0]); kernels.gram_syrk(A[1], out=G[1])
def par():
    for i in range(2):
        with torch.cuda.stream(S[i]):
            kernels.gram_syrk(A[i], out=G[i])
for f in (seq, par):
    f(); torch.cuda.synchronize()
for rep in range(3):
    for name, f in (("one stream", seq), ("two streams", par)):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name}: {dt * 1e3:.1f} ms = {2 * n * (n + 1) * p / dt / 1e12:.1f} TF", flush=True)
