"""numpy prototype of the band -> tridiagonal bulge chasing (wavefront-scheduled tasks) and of the
blocked application of its reflectors (design aid for a two-stage tridiagonalisation)."""
import numpy as np


def house(x):
    """(v, tau, beta) with v[0] = 1, (I - tau v v^T) x = beta e1."""
    alpha = x[0]
    ss = float(x[1:] @ x[1:])
    if ss == 0.0:
        v = np.zeros_like(x); v[0] = 1.0
        return v, 0.0, alpha
    beta = -np.copysign(np.sqrt(alpha * alpha + ss), alpha)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta); v[0] = 1.0
    return v, tau, beta


def sb2st(A, b, order="wavefront", rng=None):
    """A: full symmetric matrix with bandwidth b (|i-j| <= b). Returns (d, e, refl) where
    refl[(s,k)] = (c0, v, tau)."""
    A = A.copy()
    n = A.shape[0]
    refl = {}
    tasks = []
    for s in range(n - 2):
        k = 0
        while s + 1 + k * b < n:
            if k > 0 and s + 1 + k * b >= n:
                break
            tasks.append((s, k))
            k += 1
    if order == "wavefront":
        tasks.sort(key=lambda sk: (2 * sk[0] + sk[1], -sk[0]))  # same t: arbitrary (here reverse sweep order)
    for (s, k) in tasks:
        c0 = s + 1 + k * b
        L = min(b, n - c0)
        if L <= 0:
            continue
        R = slice(c0, c0 + L)
        if k == 0:
            if L < 2:
                refl[(s, k)] = (c0, np.ones(L), 0.0)
                continue
            v, tau, beta = house(A[R, s].copy())
            A[R, s] = 0.0; A[c0, s] = beta
            A[s, R] = A[R, s]
        else:
            # (i) right-apply H_{k-1} to E = A[R, Rprev]
            pc0, pv, ptau = refl[(s, k - 1)]
            Rp = slice(pc0, pc0 + len(pv))
            E = A[R, Rp]
            E -= ptau * np.outer(E @ pv, pv)
            # (ii) new reflector from E's first column
            if L < 2:
                v, tau = np.ones(L), 0.0
            else:
                v, tau, beta = house(E[:, 0].copy())
                E -= tau * np.outer(v, v @ E)
                E[1:, 0] = 0.0
            A[R, Rp] = E
            A[Rp, R] = E.T
        # (iii) two-sided on the diagonal block
        D = A[R, R]
        p = tau * (D @ v)
        w = p - 0.5 * tau * (p @ v) * v
        D -= np.outer(v, w) + np.outer(w, v)
        A[R, R] = D
        refl[(s, k)] = (c0, v, tau)
    d = np.diag(A).copy()
    e = np.diag(A, -1).copy()
    off = A - np.diag(d) - np.diag(e, 1) - np.diag(e, -1)
    return d, e, refl, np.abs(off).max()


def apply_q2(Zt, refl, n, b, w):
    """Zt <- Zt * Q2^T  with Q2 = product of H(s,k) in generation order; groups of w sweeps."""
    Zt = Zt.copy()
    smax = n - 3
    groups = [(g0, min(g0 + w, smax + 1)) for g0 in range(0, smax + 1, w)]
    for (g0, g1) in reversed(groups):
        kmax = max(k for (s, k) in refl if g0 <= s < g1)
        for k in range(0, kmax + 1):
            # block reflector of sweeps g0..g1-1 at level k, applied in reverse generation order:
            # Zt <- Zt * H(g1-1,k) * ... * H(g0,k)
            for s in range(g1 - 1, g0 - 1, -1):
                if (s, k) not in refl:
                    continue
                c0, v, tau = refl[(s, k)]
                R = slice(c0, c0 + len(v))
                Zt[:, R] -= tau * np.outer(Zt[:, R] @ v, v)
    return Zt


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n, b in [(40, 4), (97, 8), (150, 16), (64, 64 // 2)]:
        M = rng.standard_normal((n, n)); M = (M + M.T) / 2
        band = np.triu(np.tril(M, b), -b)
        ref = np.linalg.eigvalsh(band)
        for order in ["sequential", "wavefront"]:
            d, e, refl, off = sb2st(band, b, order)
            T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
            w_ = np.linalg.eigvalsh(T)
            print(n, b, order, "eig err", np.abs(w_ - ref).max(), "offband", off)
        # eigenvectors: band = Q2 T Q2^T  -> eigvecs(band) = Q2 eigvecs(T)
        wT, ZT = np.linalg.eigh(T)
        for wgrp in [1, 3, b]:
            Zt = apply_q2(ZT.T.copy(), refl, n, b, wgrp)
            Z = Zt.T
            print("   w=", wgrp, "resid", np.abs(band @ Z - Z * wT).max(), "orth", np.abs(Z.T @ Z - np.eye(n)).max())


def apply_q2_wavefront(Zt, refl, n, b, w, rng):
    """Blocks (group g, level k) applied in wavefront order tau = G + k (G = reverse group index),
    random order inside a wavefront step: checks the independence claim used by the GPU kernel."""
    Zt = Zt.copy()
    smax = n - 3
    groups = [(g0, min(g0 + w, smax + 1)) for g0 in range(0, smax + 1, w)]
    ng = len(groups)
    blocks = []
    for gi, (g0, g1) in enumerate(groups):
        ks = sorted({k for (s, k) in refl if g0 <= s < g1})
        for k in ks:
            blocks.append((ng - 1 - gi + k, gi, k))
    steps = {}
    for tau, gi, k in blocks:
        steps.setdefault(tau, []).append((gi, k))
    for tau in sorted(steps):
        lst = steps[tau]
        rng.shuffle(lst)
        for gi, k in lst:
            g0, g1 = groups[gi]
            for s in range(g1 - 1, g0 - 1, -1):
                if (s, k) not in refl:
                    continue
                c0, v, tv = refl[(s, k)]
                R = slice(c0, c0 + len(v))
                Zt[:, R] -= tv * np.outer(Zt[:, R] @ v, v)
    return Zt


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for n, b in [(97, 8), (150, 16), (200, 8)]:
        M = rng.standard_normal((n, n)); M = (M + M.T) / 2
        band = np.triu(np.tril(M, b), -b)
        d, e, refl, off = sb2st(band, b, "wavefront")
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        wT, ZT = np.linalg.eigh(T)
        for wgrp in [b, b // 2]:
            Z = apply_q2_wavefront(ZT.T.copy(), refl, n, b, wgrp, rng).T
            print("wavefront Q2", n, b, "w=", wgrp, "resid", np.abs(band @ Z - Z * wT).max())
