"""Time of the convolution weight and input rules on the ResNet-32 shapes of BASELINE config 4 (N = 1024, one slice) and their
error against fp64 autograd on a small batch.  Run once per setting of VIVIT_CONV_MFMA (read once per process)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, torch.nn.functional as F
from vivit_amd import kernels
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = [(3, 16, 32, 1), (16, 16, 32, 1), (16, 32, 32, 2), (32, 32, 16, 1), (32, 64, 16, 2), (64, 64, 8, 1)]
tot_w = tot_j = 0.0
for cin, cout, hw, s in shapes:
    x = torch.randn(N, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
    oh = (hw + 2 - 3) // s + 1
    M = torch.randn(1, N, cout, oh, oh, device=dev)
    def t(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    tw = t(lambda: kernels.conv2d_weight_mjp(M, x, (3, 3), (s, s), (1, 1), (1, 1)))
    tj = t(lambda: kernels.conv2d_jac_t(M, w, (hw, hw), (s, s), (1, 1), (1, 1)))
    # accuracy on 4 samples against fp64
    xs, Ms = x[:4].double(), M[:, :4].double()
    got = kernels.conv2d_weight_mjp(M[:, :4].contiguous(), x[:4].contiguous(), (3, 3), (s, s), (1, 1), (1, 1))
    xu = F.unfold(xs, (3, 3), padding=1, stride=s)
    ref = torch.einsum("vnol,nkl->vnok", Ms.flatten(3), xu).reshape(got.shape)
    ew = ((got.double() - ref).abs().max() / ref.abs().max()).item()
    gj = kernels.conv2d_jac_t(M[:, :4].contiguous(), w, (hw, hw), (s, s), (1, 1), (1, 1))
    refj = F.conv_transpose2d(Ms[0], w.double(), stride=s, padding=1, output_padding=hw - ((oh - 1) * s + 1))
    ej = ((gj[0].double() - refj).abs().max() / refj.abs().max()).item()
    flops = 2.0 * N * cout * cin * 9 * oh * oh
    mult = {(3, 16): 1, (16, 16): 10, (16, 32): 1, (32, 32): 9, (32, 64): 1, (64, 64): 9}[(cin, cout)]
    tot_w += mult * tw; tot_j += mult * tj
    print(f"{cin:3d}->{cout:3d} @{hw:2d} s{s}: weight rule {tw*1e3:7.1f} us ({flops/tw/1e9:6.1f} TF) err {ew:.1e} | input rule {tj*1e3:7.1f} us ({flops/tj/1e9:6.1f} TF) err {ej:.1e}", flush=True)
print(f"MFMA={os.environ.get('VIVIT_CONV_MFMA', '1')}: ResNet-32 totals per backward: weight rules {tot_w:.2f} ms, input rules {tot_j:.2f} ms")
