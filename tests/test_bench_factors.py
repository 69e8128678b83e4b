"""bench.py's synthetic factor generator must produce exactly what SqrtGGNExact attaches to the
parameters (checked on CPU against the stand-in backend, which is itself checked against autograd)."""
import torch
from torch import nn

import bench
from helpers import OracleBackend, set_kernel_backend
from vivit_amd import kernels
from vivit_amd.backend import SqrtGGNExact, backpack, extend


def test_bench_factors_match_backend():
    dims, batch = (7, 6, 5), 4
    facs = bench.mlp_sqrt_ggn_factors(dims, batch, torch.device("cpu"), seed=3)
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(3)
        lin1, lin2 = nn.Linear(7, 6), nn.Linear(6, 5)
        X = torch.rand(batch, 7)
    model = extend(nn.Sequential(lin1, nn.ReLU(), lin2))
    lossf = extend(nn.CrossEntropyLoss())
    set_kernel_backend(OracleBackend())
    try:
        loss = lossf(model(X), torch.zeros(batch, dtype=torch.long))
        with backpack(SqrtGGNExact()):
            loss.backward()
    finally:
        set_kernel_backend(None)
    n = 5 * batch
    expect = [lin2.weight, lin2.bias, lin1.weight, lin1.bias]
    for f, p in zip(facs, expect):
        torch.testing.assert_close(f, p.sqrt_ggn_exact.reshape(n, -1), rtol=1e-5, atol=1e-7)
    # sharding the first-layer weight over 2 ranks partitions its columns
    f0 = bench.mlp_sqrt_ggn_factors(dims, batch, torch.device("cpu"), shard=(0, 2), seed=3)
    f1 = bench.mlp_sqrt_ggn_factors(dims, batch, torch.device("cpu"), shard=(1, 2), seed=3)
    G_full = sum(f @ f.T for f in facs)
    G_sharded = sum(f @ f.T for f in f0) + sum(f @ f.T for f in f1)
    torch.testing.assert_close(G_full, G_sharded, rtol=1e-5, atol=1e-6)


def test_bench_factors_batch_shard_rows():
    """``samples=(lo, hi)`` (a data-parallel rank's shard) = the corresponding class-major rows of the full factors."""
    dims, batch, C = (7, 6, 5), 6, 5
    full = bench.mlp_sqrt_ggn_factors(dims, batch, torch.device("cpu"), seed=3)
    lo, hi = 2, 4
    part = bench.mlp_sqrt_ggn_factors(dims, batch, torch.device("cpu"), seed=3, samples=(lo, hi))
    for f, p in zip(full, part):
        expect = f.view(C, batch, -1)[:, lo:hi].reshape(C * (hi - lo), -1)
        torch.testing.assert_close(p, expect, rtol=1e-5, atol=1e-8)  # (S @ W2 is blocked differently on a slice)


def test_rank_watchdog_ends_the_survivors_when_one_rank_dies():
    """bench._watch_ranks (the bare `--gpus N` launcher): a rank that exits non-zero must not leave the others blocked in a
    collective until the driver's timeout -- they are terminated and the launcher reports the failure."""
    import subprocess
    import sys
    import time

    sleeper = "import time; time.sleep(600)"
    procs = [subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.5); sys.exit(3)"], start_new_session=True),
             subprocess.Popen([sys.executable, "-c", sleeper], start_new_session=True),
             subprocess.Popen([sys.executable, "-c", sleeper], start_new_session=True)]
    t0 = time.monotonic()
    rc = bench._watch_ranks(procs, poll_s=0.1, grace_s=5.0)
    assert rc == 3
    assert time.monotonic() - t0 < 30
    assert all(p.poll() is not None for p in procs)
    # all ranks fine -> 0
    ok = [subprocess.Popen([sys.executable, "-c", "pass"], start_new_session=True) for _ in range(2)]
    assert bench._watch_ranks(ok, poll_s=0.1) == 0
