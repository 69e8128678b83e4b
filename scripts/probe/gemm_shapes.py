"""Which products of one symeig (n from argv) go through the 256-tile kernels (VIVIT_GEMM_DEBUG=1 prints the plan)."""
import os, sys, collections, subprocess
if os.environ.get("CHILD"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import torch
    from vivit_amd import kernels
    n = int(sys.argv[1])
    V = torch.randn(n, n // 2, device="cuda")
    G = kernels.gram_syrk(V)
    torch.cuda.synchronize()
    print("MARK", file=sys.stderr, flush=True)
    kernels.symeig(G, eigenvectors=True, overwrite=True)
    torch.cuda.synchronize()
else:
    env = dict(os.environ, CHILD="1", VIVIT_GEMM_DEBUG="1")
    out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True).stderr
    out = out.split("MARK", 1)[1]
    c = collections.Counter()
    for line in out.splitlines():
        if line.startswith("gemm256:"):
            parts = dict(p.split("=") for p in line.split()[1:])
            key = (int(parts["M"]) // 1024, int(parts["N"]) // 1024, int(parts["K"]) // 512, parts["syrk"], parts["ksplit"], parts["lay"], parts["bxws"])
            c[key] += 1
    for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:40]:
        print("M~%dk N~%dk K~%d*512 syrk=%s ksplit=%s lay=%s bxws=%s : %d calls" % (*k, v))
