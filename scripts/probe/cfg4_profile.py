"""Config 4 (ResNet-32, N = 1024, MC mc = 1): where the backward pass with the factor extension spends its GPU time
(torch.profiler, steady state) against the plain backward."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_configs as bc  # noqa: E402
import vivit_amd  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
N = 1024
model = bc.resnet32(100).to(dev)
X, y = torch.rand(N, 3, 32, 32, device=dev), torch.randint(0, 100, (N,), device=dev)
with torch.no_grad():
    idx = torch.multinomial(model(X).softmax(1), 1, replacement=True)
    samples = torch.nn.functional.one_hot(idx.t(), 100).float()
pb = bc._Problem(model, X, y, samples)
comp = vivit_amd.EigvalshComputation(mc_samples=1)
ext = [comp.get_extension()]


def run(with_ext):
    pb.backward(ext if with_ext else ())
    torch.cuda.synchronize()


for w in (False, True):
    for _ in range(3):
        run(w)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            run(w)
    print("==== with factor extension" if w else "==== plain backward")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
