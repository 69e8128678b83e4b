"""numpy prototype of the divide-and-conquer tridiagonal eigensolver that symeig_large.hip
implements (debug / design aid; not product, not oracle).

Mirrors the GPU data flow: eigenvector matrix stored TRANSPOSED (row = eigenvector), fp32 GEMMs,
fp64 secular-equation solve, rank-based sort, deflation scan, gather of non-deflated rows.
"""
import numpy as np

f32 = np.float32
EPS32 = float(np.finfo(np.float32).eps) / 2  # 2^-24


def leaf_eig(d, e):
    T = np.diag(d.astype(np.float64)) + np.diag(e.astype(np.float64), 1) + np.diag(e.astype(np.float64), -1)
    w, Q = np.linalg.eigh(T)
    return w.astype(np.float32), Q.T.astype(np.float32).copy()  # rows = eigenvectors


def secular_roots(dk, z2, rho, maxit=200):
    """Roots of 1 + rho * sum z2_j / (dk_j - lam) = 0, one per interval; fp64.
    Returns (origin index per root, mu) with lam_i = dk[origin_i] + mu_i."""
    k = len(dk)
    org = np.zeros(k, np.int64)
    mu = np.zeros(k, np.float64)
    znorm2 = z2.sum()
    for i in range(k):
        lo_pole = dk[i]
        hi_pole = dk[i + 1] if i + 1 < k else dk[k - 1] + rho * znorm2
        # choose origin: evaluate at midpoint
        mid = 0.5 * (lo_pole + hi_pole)
        if i + 1 < k:
            fmid = 1.0 + rho * np.sum(z2 / (dk - mid))
            o = i if fmid >= 0 else i + 1
        else:
            o = i
        delta = dk - dk[o]
        # g(mu) = 1 + rho sum z2/(delta - mu), increasing in mu on the interval
        if o == i:
            a, b = 0.0, (hi_pole - dk[o])  # root in (0, b)
            # geometric search downward from b/2 until g < 0
            x = 0.5 * b
            if i + 1 == k:
                x = b  # g(b) >= 0 at the upper bound for the last root
            gx = 1.0 + rho * np.sum(z2 / (delta - x))
            hi = None
            while gx > 0:
                hi = x
                x *= 0.5
                if x == 0.0:
                    break
                gx = 1.0 + rho * np.sum(z2 / (delta - x))
            lo = x
            if hi is None:
                hi = b
        else:
            a, b = (lo_pole - dk[o]), 0.0  # root in (a, 0), a < 0
            x = 0.5 * a
            gx = 1.0 + rho * np.sum(z2 / (delta - x))
            lo = None
            while gx < 0:
                lo = x
                x *= 0.5
                if x == 0.0:
                    break
                gx = 1.0 + rho * np.sum(z2 / (delta - x))
            hi = x
            if lo is None:
                lo = a
        for _ in range(maxit):
            m = 0.5 * (lo + hi)
            if m == lo or m == hi:
                break
            gm = 1.0 + rho * np.sum(z2 / (delta - m))
            if gm > 0:
                hi = m
            else:
                lo = m
        org[i] = o
        mu[i] = 0.5 * (lo + hi)
    return org, mu


def merge(d1, Q1t, d2, Q2t, rho_signed, stats):
    n1, n2 = len(d1), len(d2)
    n = n1 + n2
    sgn = 1.0 if rho_signed >= 0 else -1.0
    rho = abs(float(rho_signed))
    d = np.concatenate([d1, d2]).astype(np.float64)
    z = np.concatenate([Q1t[:, -1].astype(np.float64), sgn * Q2t[:, 0].astype(np.float64)])
    # Qt block-diagonal: rows 0..n1-1 live in cols 0..n1-1
    Qt = np.zeros((n, n), np.float32)
    Qt[:n1, :n1] = Q1t
    Qt[n1:, n1:] = Q2t
    # normalise z (||z||^2 = 2)
    zn = np.sqrt(np.sum(z * z))
    z = z / zn
    rho = rho * zn * zn
    order = np.argsort(d, kind="stable")
    tol = 8.0 * EPS32 * max(np.abs(d).max(), np.abs(z).max())
    # deflation scan in sorted order
    nd = []  # non-deflated physical indices in sorted order
    defl = []
    rots = []
    if rho * np.abs(z).max() <= tol:
        # everything deflates
        keep_idx = np.array([], np.int64)
        defl = list(order)
    else:
        prev = -1
        for idx in order:
            if rho * abs(z[idx]) <= tol:
                defl.append(idx)
                continue
            if prev >= 0:
                s_ = z[prev]
                c_ = z[idx]
                tau = np.hypot(c_, s_)
                t = d[idx] - d[prev]
                c_ /= tau
                s_ = -s_ / tau
                if abs(t * c_ * s_) <= tol:
                    # deflate prev: rotate rows (prev, idx)
                    z[idx] = tau
                    z[prev] = 0.0
                    rots.append((prev, idx, c_, s_))
                    dp = d[prev] * c_ * c_ + d[idx] * s_ * s_
                    di = d[prev] * s_ * s_ + d[idx] * c_ * c_
                    d[prev], d[idx] = dp, di
                    defl.append(prev)
                    nd.pop()  # prev was tentatively non-deflated
                    # note: d[prev] changed; deflated values get re-sorted at the next level
            nd.append(idx)
            prev = idx
    for (p, q, c_, s_) in rots:
        rp = Qt[p].copy(); rq = Qt[q].copy()
        Qt[p] = f32(c_) * rp + f32(s_) * rq
        Qt[q] = f32(-s_) * rp + f32(c_) * rq
    k = len(nd)
    stats.append((n, k))
    out_d = np.zeros(n, np.float32)
    out_Qt = np.zeros((n, n), np.float32)
    if k > 0:
        nd = np.array(nd)
        dk = d[nd]
        zk = z[nd]
        assert np.all(np.diff(dk) >= 0)
        org, mu = secular_roots(dk, zk * zk, rho)
        lam = dk[org] + mu
        # Loewner: zhat_j^2 = prod_i (lam_i - d_j) / (rho * prod_{i != j} (d_i - d_j))
        zhat = np.zeros(k)
        for j in range(k):
            num = (dk[org] - dk[j]) + mu  # lam_i - d_j, accurate
            den = dk - dk[j]
            prod = num[j] / rho
            for i in range(k):
                if i != j:
                    prod *= num[i] / den[i]
            zhat[j] = np.sign(zk[j]) * np.sqrt(abs(prod))
        U = np.zeros((k, k))  # U[j, i]
        for i in range(k):
            col = zhat / ((dk - dk[org[i]]) - mu[i])
            U[:, i] = col / np.sqrt(np.sum(col * col))
        G = Qt[nd]  # gather [k, n]
        out_Qt[:k] = (U.astype(np.float32).T @ G)
        out_d[:k] = lam.astype(np.float32)
    for t, idx in enumerate(defl):
        out_Qt[k + t] = Qt[idx]
        out_d[k + t] = f32(d[idx])
    return out_d, out_Qt


def dc(d, e, leaf=32, stats=None):
    d = d.astype(np.float32).copy()
    e = e.astype(np.float32).copy()
    n = len(d)
    if stats is None:
        stats = []

    def rec(lo, hi):
        if hi - lo <= leaf:
            return leaf_eig(d[lo:hi], e[lo:hi - 1])
        mid = (lo + hi) // 2
        rho = e[mid - 1]
        d[mid - 1] -= abs(rho)
        d[mid] -= abs(rho)
        d1, Q1 = rec(lo, mid)
        d2, Q2 = rec(mid, hi)
        return merge(d1, Q1, d2, Q2, rho, stats)

    w, Qt = rec(0, n)
    order = np.argsort(w, kind="stable")
    return w[order], Qt[order], stats


if __name__ == "__main__":
    import sys
    sys.path.insert(0, "scripts")
    from ql_proto import tridiag
    rng = np.random.default_rng(0)
    for n, kind in [(200, "dense"), (300, "lowrank"), (400, "gramdecay"), (256, "wilk"), (300, "clustered")]:
        if kind == "dense":
            M = rng.standard_normal((n, n)); S = (M + M.T) / 2
            d, e = tridiag(S.astype(np.float32)); e = e[:-1]
        elif kind == "lowrank":
            V = rng.standard_normal((n, n // 3)); S = V @ V.T
            d, e = tridiag(S.astype(np.float32)); e = e[:-1]
        elif kind == "gramdecay":
            V = rng.standard_normal((n, 2 * n)) * (0.9 ** np.arange(2 * n))[None, :]; S = V @ V.T
            d, e = tridiag(S.astype(np.float32)); e = e[:-1]
        elif kind == "wilk":
            d = np.abs(np.arange(n) - n // 2).astype(np.float32); e = np.ones(n - 1, np.float32)
        else:
            d = np.repeat(np.array([1.0, 2.0, 3.0], np.float32), n // 3); e = np.full(n - 1, 1e-4, np.float32)
        T = np.diag(d.astype(np.float64)) + np.diag(e.astype(np.float64), 1) + np.diag(e.astype(np.float64), -1)
        ref = np.linalg.eigvalsh(T)
        w, Qt, stats = dc(d, e)
        sc = np.abs(ref).max()
        Q = Qt.T.astype(np.float64)
        orth = np.abs(Q.T @ Q - np.eye(n)).max()
        res = np.abs(T @ Q - Q * w[None, :].astype(np.float64)).max() / sc
        print(f"{kind:10s} n={n} eval err {np.abs(w - ref).max()/sc:.2e} orth {orth:.2e} resid {res:.2e} top-merge (n,k)={stats[-1]}")
