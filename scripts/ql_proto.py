"""float32 emulation of symeig_small.hip's recurrences (debug aid; not product, not oracle)."""
import numpy as np
import torch

f32 = np.float32
EPS = f32(5.9604645e-8)


def tridiag(A):
    A = A.astype(np.float32).copy()
    n = A.shape[0]
    d = np.zeros(n, np.float32); e = np.zeros(n, np.float32)
    for j in range(n - 2):
        x = A[j + 1:, j].copy()
        ss = f32(np.sum(x[1:] * x[1:], dtype=np.float32))
        alpha = x[0]
        tj = f32(0); ej = alpha
        if ss > 0:
            beta = -np.copysign(np.sqrt(alpha * alpha + ss, dtype=np.float32), alpha)
            tj = (beta - alpha) / beta
            scal = f32(1) / (alpha - beta)
            ej = beta
            x[1:] *= scal
        x[0] = 1
        d[j] = A[j, j]; e[j] = ej
        if tj != 0:
            A22 = A[j + 1:, j + 1:]
            p = tj * (A22 @ x)
            a2 = f32(-0.5) * tj * f32(p @ x)
            w = p + a2 * x
            A22 -= np.outer(x, w) + np.outer(w, x)
    d[n - 2] = A[n - 2, n - 2]; e[n - 2] = A[n - 1, n - 2]; d[n - 1] = A[n - 1, n - 1]
    return d, e


def ql(d, e, maxit=60, mode="rel"):
    d = d.copy(); e = e.copy(); n = len(d)
    nfail = 0; total = 0
    tn = f32(np.max(np.abs(d) + np.abs(e)))
    for l in range(n):
        it = 0
        while True:
            m = l
            while m < n - 1:
                dd = abs(d[m]) + abs(d[m + 1]); ae = abs(e[m])
                if mode == "rel":
                    if ae <= EPS * dd or ae < 1e-37: break
                else:
                    if ae <= EPS * dd or ae <= EPS * tn * f32(0.5): break
                m += 1
            if m == l: break
            if it >= maxit:
                nfail += 1; break
            it += 1; total += 1
            g = (d[l + 1] - d[l]) / (f32(2) * e[l])
            r = np.sqrt(g * g + f32(1), dtype=np.float32)
            g = d[m] - d[l] + e[l] / (g + np.copysign(r, g))
            s = f32(1); c = f32(1); p = f32(0)
            i = m - 1; broke = False
            while i >= l:
                f = s * e[i]; b = c * e[i]
                r = np.sqrt(f * f + g * g, dtype=np.float32)
                e[i + 1] = r
                if r == 0:
                    d[i + 1] -= p; e[m] = 0; broke = True; break
                s = f / r; c = g / r
                g = d[i + 1] - p
                r = (d[i] - g) * s + f32(2) * c * b
                p = s * r
                d[i + 1] = g + p
                g = c * r - b
                i -= 1
            if broke: continue
            d[l] -= p; e[l] = g; e[m] = 0
    return np.sort(d), nfail, total


if __name__ == "__main__":
    np.seterr(all="ignore")
    for n in [15, 33, 64, 100, 128, 192]:
        g = torch.Generator().manual_seed(n * 31 + 0)
        r = max(1, n // 3)
        V = torch.randn(n, r, generator=g)
        S = (V @ V.T).numpy()
        d, e = tridiag(S)
        ref = np.linalg.eigvalsh(S.astype(np.float64))
        for mode in ["rel", "abs"]:
            w, nfail, total = ql(d, e, mode=mode)
            print(n, mode, "nfail", nfail, "iters", total, "maxerr", np.abs(w - ref).max() / np.abs(ref).max())
