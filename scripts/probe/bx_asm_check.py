"""The hand-scheduled K loop of gemm256_bx_kernel<6> (VIVIT_BX_ASM=1, csrc/bx_kloop_asm.inc) against the C++ loop (VIVIT_BX_ASM=0):
bit-identical outputs on a set of shapes (full / edge tiles, one and several accumulation chains, SYRK with mirror, NT products
with beta), then the headline-shaped SYRK timed with both, interleaved, on the same box.

    python scripts/probe/bx_asm_check.py [--lib path/to/lib.so] [--time]
"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(do_time):
    sys.path.insert(0, ROOT)
    import torch
    from vivit_amd import kernels

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    out = []
    for n, p in ((1536, 40000), (3000, 9000), (5120, 16384), (777, 70000)):
        A = torch.randn(n, p, device=dev, generator=g)
        G = kernels.gram_syrk(A)
        out.append(("syrk", n, p, hashlib.sha256(G.cpu().numpy().tobytes()).hexdigest()[:16]))
    for m, n, k in ((2048, 4096, 2048), (1000, 3000, 5000), (4096, 4096, 512)):
        A = torch.randn(m, k, device=dev, generator=g)
        B = torch.randn(n, k, device=dev, generator=g)
        C = torch.randn(m, n, device=dev, generator=g)
        kernels.gemm_nt(A, B, out=C, alpha=-1.0, beta=1.0)
        out.append(("gemm_nt", m, n, k, hashlib.sha256(C.cpu().numpy().tobytes()).hexdigest()[:16]))
    for row in out:
        print("HASH", *row, flush=True)
    if do_time:
        n, p = 40960, 131072
        G = torch.empty(n, n, device=dev)
        A = torch.randn(n, p, device=dev, generator=g)
        for kind in ("randn", "half zeros"):
            if kind == "half zeros":
                A.mul_((torch.rand(n, p // 784 + 1, device=dev, generator=g) < 0.5).repeat_interleave(784, 1)[:, :p])
            kernels.gram_syrk(A, out=G)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                kernels.gram_syrk(A, out=G)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t = sorted(ts)[1]
            print(f"TIME {kind}: {t * 1e3:.1f} ms = {n * (n + 1) * p / t / 1e12:.1f} TF", flush=True)
        print("HASH", "headline-shaped", hashlib.sha256(G[:4096].cpu().numpy().tobytes()).hexdigest()[:16], flush=True)


MODES = ("0", "1")   # VIVIT_BX_ASM: C++ loop | asm loop


def main():
    if "--child" in sys.argv:
        return child("--time" in sys.argv)
    env = dict(os.environ)
    if "--lib" in sys.argv:
        env["VIVIT_HIP_LIB"] = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
        env["VIVIT_HIP_ALLOW_STALE"] = "1"
    reps = 2 if "--time" in sys.argv else 1
    hashes = {}
    for rep in range(reps):
        for asm in MODES:
            e = dict(env, VIVIT_BX_ASM=asm)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + (["--time"] if "--time" in sys.argv else []),
                               env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            lines = [l for l in r.stdout.splitlines() if l.startswith(("HASH", "TIME"))]
            if r.returncode != 0:
                print(r.stdout[-3000:])
                raise SystemExit(f"child failed (VIVIT_BX_ASM={asm})")
            hashes.setdefault(asm, [l for l in lines if l.startswith("HASH")])
            for l in lines:
                if l.startswith("TIME"):
                    print(f"asm={asm} rep {rep}: {l}", flush=True)
    same = all(hashes[m] == hashes[MODES[0]] for m in MODES)
    for row in zip(*[hashes[m] for m in MODES]):
        print(("same " if len(set(row)) == 1 else "DIFF ") + "   |   ".join(sorted(set(row))))
    print("bit-identical:", same, "modes", MODES)
    raise SystemExit(0 if same else 1)


if __name__ == "__main__":
    main()
