"""Timing of the bf16-pipe tile kernel with the C++ K loop (VIVIT_BX_ASM=0) and the asm K loop (=1) on the OTHER shapes it serves:
small-output / deep-K Gram matrices (split-K launches), mid-size SYRKs, Q1-like and trailing-update-like products.
usage: python scripts/probe/bx_asm_shapes.py   (spawns one child per mode)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    for n, p in ((1280, 401408), (1024, 470016), (5120, 65536), (20480, 48128), (10240, 407056)):
        A = torch.randn(n, p, device=dev, generator=g)
        G = torch.empty(n, n, device=dev)
        t = timed(lambda: kernels.gram_syrk(A, out=G))
        print(f"RES syrk n={n} p={p}: {t * 1e3:.2f} ms = {n * (n + 1) * p / t / 1e12:.1f} TF", flush=True)
        del A, G
    for m, n, k, tag in ((40960, 2048, 20480, "Q1 product 1"), (40960, 20480, 2048, "Q1 product 3"), (20480, 20480, 1024, "trailing-update-like NT")):
        A = torch.randn(m, k, device=dev, generator=g)
        B = torch.randn(n, k, device=dev, generator=g)
        C = torch.randn(m, n, device=dev, generator=g)
        t = timed(lambda: kernels.gemm_nt(A, B, out=C, alpha=-1.0, beta=1.0), reps=3)
        print(f"RES gemm_nt {tag} {m}x{n}x{k}: {t * 1e3:.2f} ms = {2.0 * m * n * k / t / 1e12:.1f} TF", flush=True)
        del A, B, C


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for rep in range(2):
            for asm in ("0", "1"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, VIVIT_BX_ASM=asm),
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                for l in r.stdout.splitlines():
                    if l.startswith("RES"):
                        print(f"asm={asm} rep {rep}: {l[4:]}", flush=True)
                if r.returncode != 0:
                    print(r.stdout[-2000:])
                    raise SystemExit(1)
