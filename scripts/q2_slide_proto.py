"""Lane-level numpy emulation of the sliding-window Q2 back-transformation (csrc/q2slide.hip): design aid that pins
the index algebra of the kernel before it runs on the GPU.

  * reflectors of the bulge chase as sb2st.hip stores them: v(s, k) at R2[s][c0 .. c0 + L), tau2[s][k], c0 = s + 1 + 64 k
  * `prepare_block`: the per-block image = 78 A-operand fragments (1 KB each: 64 lanes x 8 bf16) in consumption order,
    three bf16 pieces (hi, mid, lo) per fragment triple
  * `apply_wave`: one wave = 16 rows of Zt; the window of three 64-column units lives in registers
    (lane (n16, kq) holds S[row n16][16 q + 4 kq + e]); W2^T = (T V) S^T and U^T = V^T W2^T on v_mfma_f32_16x16x32_bf16 with
    six partial products; accumulators feed the next product as B operands without lane movement
  * `walk`: pass K (levels 2K, 2K + 1), groups from the last one that has level 2K down to 0; the window slides left by one
    unit per group

Checked against the sequential application of every reflector (scripts/sb2st_proto.py).
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from sb2st_proto import sb2st  # noqa: E402

NB = 64
QW = 64


# ---------------------------------------------------------------- bf16 pieces (round to nearest even, as (__bf16) casts)
def bf16_round(x):
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    hi = bf16_round(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_round(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_round(r2)
    return hi, mid, lo


# ---------------------------------------------------------------- v_mfma_f32_16x16x32_bf16, documented lane maps
LANE = np.arange(64)
L16, LKQ = LANE & 15, LANE >> 4


def kcol(ks, kq, j):
    """k index of fragment element j of lanes kq in k-step ks (the permuted order both operands use)"""
    return 32 * ks + 16 * (j >> 2) + 4 * kq + (j & 3)


def mfma(Af, Bf, C):
    """Af, Bf: [64 lanes][8]; C: [64][4].  A[row l&15][k = 8 (l>>4) + j], B[k = 8 (l>>4) + j][col l&15];
    D[row 4 (l>>4) + e][col l&15] in element e."""
    A = np.zeros((16, 32), dtype=np.float64)
    B = np.zeros((32, 16), dtype=np.float64)
    for ln in range(64):
        for j in range(8):
            A[ln & 15, 8 * (ln >> 4) + j] = Af[ln, j]
            B[8 * (ln >> 4) + j, ln & 15] = Bf[ln, j]
    D = A @ B
    out = C.copy()
    for ln in range(64):
        for e in range(4):
            out[ln, e] = np.float32(out[ln, e] + D[4 * (ln >> 4) + e, ln & 15])
    return out


def mfma6(Ap, Bp, C):
    """six partial products, smallest first; Ap, Bp = (hi, mid, lo)"""
    for a, b in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
        C = mfma(Ap[a], Bp[b], C)
    return C


# ---------------------------------------------------------------- fragment lists (consumption order)
W2_LIST = [(ks, ta) for ks in range(4) for ta in range(4) if 32 * ks + 31 >= 16 * ta + 1]  # (T V)[t'][w] != 0 needs w >= t' + 1
U_LIST = [(wt, kt) for wt in range(8) for kt in range(2)
          if (32 * kt <= 16 * wt + 14) and (32 * kt + 31 >= 16 * wt - 64)]                 # V[t][w] != 0 needs w-64 <= t <= w-1
assert len(W2_LIST) == 14 and len(U_LIST) == 12, (len(W2_LIST), len(U_LIST))
NFRAG = 3 * (len(W2_LIST) + len(U_LIST))


def block_V(R2, tau2, n, g, k):
    """Vw[t][w] (window column w holds reflector component i = w - 1) and taus of block (group g, level k)"""
    g0 = g * QW
    c_start = g0 + 1 + k * NB
    Vw = np.zeros((QW, 128), dtype=np.float32)
    taus = np.zeros(QW, dtype=np.float32)
    for t in range(QW):
        s, c0 = g0 + t, c_start + t
        if s > n - 3 or c0 >= n:
            continue
        L = min(NB, n - c0)
        Vw[t, t + 1:t + 1 + L] = R2[s, c0:c0 + L]
        taus[t] = tau2[s, k]
    return Vw, taus


def prepare_block(R2, tau2, n, g, k, exact=False):
    """image [NFRAG][64][8] (float32 holding bf16-representable values; exact=True: hi = value, mid = lo = 0)"""
    Vw, taus = block_V(R2, tau2, n, g, k)
    S = (Vw.astype(np.float64) @ Vw.astype(np.float64).T).astype(np.float32)
    T = np.zeros((QW, QW), dtype=np.float32)
    for j in range(QW):  # tfactor_column
        T[j, j] = taus[j]
        for i in range(j - 1, -1, -1):
            T[i, j] = -taus[i] * np.float32(np.dot(S[i, i + 1:j + 1].astype(np.float64), T[i + 1:j + 1, j].astype(np.float64)))
    TV = (np.triu(T).astype(np.float64) @ Vw.astype(np.float64)).astype(np.float32)
    img = np.zeros((NFRAG, 64, 8), dtype=np.float32)
    f = 0
    for ks, ta in W2_LIST:
        vals = np.zeros((64, 8), dtype=np.float32)
        for ln in range(64):
            for j in range(8):
                vals[ln, j] = TV[16 * ta + (ln & 15), kcol(ks, ln >> 4, j)]
        pcs = (vals, np.zeros_like(vals), np.zeros_like(vals)) if exact else split3(vals)
        for p in range(3):
            img[f + p] = pcs[p]
        f += 3
    for wt, kt in U_LIST:
        vals = np.zeros((64, 8), dtype=np.float32)
        for ln in range(64):
            for j in range(8):
                vals[ln, j] = Vw[kcol(kt, ln >> 4, j), 16 * wt + (ln & 15)]
        pcs = (vals, np.zeros_like(vals), np.zeros_like(vals)) if exact else split3(vals)
        for p in range(3):
            img[f + p] = pcs[p]
        f += 3
    assert f == NFRAG
    return img


def apply_block(sw, Q0, img, exact=False):
    """sw: [12][64 lanes][4] window registers (float4 per lane); block on the float4s Q0 .. Q0 + 7"""
    acc2 = [np.zeros((64, 4), dtype=np.float32) for _ in range(4)]
    f = 0
    cur_ks, Bp = -1, None
    for ks, ta in W2_LIST:
        if ks != cur_ks:
            vals = np.concatenate([sw[Q0 + 2 * ks], sw[Q0 + 2 * ks + 1]], axis=1)  # [64][8]: j < 4 from the first float4
            Bp = (vals, np.zeros_like(vals), np.zeros_like(vals)) if exact else split3(vals)
            cur_ks = ks
        acc2[ta] = mfma6((img[f], img[f + 1], img[f + 2]), Bp, acc2[ta])
        f += 3
    Wp = []
    for kt in range(2):
        vals = np.concatenate([acc2[2 * kt], acc2[2 * kt + 1]], axis=1)
        Wp.append((vals, np.zeros_like(vals), np.zeros_like(vals)) if exact else split3(vals))
    u, cur_wt = None, -1
    for idx, (wt, kt) in enumerate(U_LIST):
        if wt != cur_wt:
            u = np.zeros((64, 4), dtype=np.float32)
            cur_wt = wt
        u = mfma6((img[f], img[f + 1], img[f + 2]), Wp[kt], u)
        f += 3
        if idx + 1 == len(U_LIST) or U_LIST[idx + 1][0] != wt:
            sw[Q0 + wt] = (sw[Q0 + wt] - u).astype(np.float32)
    assert f == NFRAG


def passes(n):
    """[(K, gmax)]: pass K handles levels 2K, 2K + 1 of groups gmax .. 0 (level 2K + 1 exists for g < gmax only)"""
    out = []
    K = 0
    while n - 2 - 128 * K >= 0:
        out.append((K, (n - 2 - 128 * K) // 64))
        K += 1
    return out


def load_unit(Zt, rows, u, n):
    """float4s of unit u for one wave: [4][64][4]"""
    out = np.zeros((4, 64, 4), dtype=np.float32)
    for q in range(4):
        for ln in range(64):
            c = 64 * u + 16 * q + 4 * (ln >> 4)
            r = rows[ln & 15]
            if r is not None and c < n:
                out[q, ln] = Zt[r, c:c + 4]
    return out


def store_unit(Zt, rows, u, n, regs):
    for q in range(4):
        for ln in range(64):
            c = 64 * u + 16 * q + 4 * (ln >> 4)
            r = rows[ln & 15]
            if r is not None and c < n:
                Zt[r, c:c + 4] = regs[q, ln]


def walk_wave(Zt, rows, R2, tau2, n, images, exact):
    """One wave's 16 rows through all passes.  images[(g, k)] -> image"""
    for K, gmax in passes(n):
        g = gmax
        sw = np.zeros((12, 64, 4), dtype=np.float32)
        for u3 in range(3):
            sw[4 * u3:4 * u3 + 4] = load_unit(Zt, rows, g + 2 * K + u3, n)
        while g >= 0:
            nxt = load_unit(Zt, rows, g - 1 + 2 * K, n) if g > 0 else None
            apply_block(sw, 0, images[(g, 2 * K)], exact)
            if g < gmax:
                apply_block(sw, 4, images[(g, 2 * K + 1)], exact)
            store_unit(Zt, rows, g + 2 * K + 2, n, sw[8:12])
            sw[8:12] = sw[4:8]
            sw[4:8] = sw[0:4]
            if nxt is not None:
                sw[0:4] = nxt
            g -= 1
        # after the last group (g = 0): units 2K (now in slot 1) and 2K + 1 (slot 2) are still in registers
        store_unit(Zt, rows, 2 * K, n, sw[4:8])
        store_unit(Zt, rows, 2 * K + 1, n, sw[8:12])


def reference(Zt, refl, n):
    Zt = Zt.astype(np.float64).copy()
    smax = n - 3
    for s in range(smax, -1, -1):  # reverse generation order: last sweep first, within a sweep ... see sb2st_proto.apply_q2
        pass
    groups = [(g0, min(g0 + QW, smax + 1)) for g0 in range(0, smax + 1, QW)]
    for (g0, g1) in reversed(groups):
        kmax = max(k for (s, k) in refl if g0 <= s < g1)
        for k in range(0, kmax + 1):
            for s in range(g1 - 1, g0 - 1, -1):
                if (s, k) not in refl:
                    continue
                c0, v, tau = refl[(s, k)]
                Rr = slice(c0, c0 + len(v))
                Zt[:, Rr] -= tau * np.outer(Zt[:, Rr] @ v, v)
    return Zt


def main():
    rng = np.random.default_rng(0)
    for n in (132, 200, 324):
        M = rng.standard_normal((n, n))
        M = (M + M.T) / 2
        band = np.triu(np.tril(M, NB), -NB)
        d, e, refl, off = sb2st(band, NB, "wavefront")
        R2 = np.zeros((n, n), dtype=np.float32)
        nk = n // NB + 2
        tau2 = np.zeros((n, nk), dtype=np.float32)
        for (s, k), (c0, v, tau) in refl.items():
            R2[s, c0:c0 + len(v)] = v
            tau2[s, k] = tau
        refl32 = {key: (c0, R2[key[0], c0:c0 + len(v)].astype(np.float64), float(tau2[key[0], key[1]]))
                  for key, (c0, v, tau) in refl.items()}
        nrows = 16
        Z0 = rng.standard_normal((nrows, n)).astype(np.float32) / np.sqrt(n)
        ref = reference(Z0, refl32, n)
        for exact in (True, False):
            images = {}
            for K, gmax in passes(n):
                for g in range(gmax, -1, -1):
                    images[(g, 2 * K)] = prepare_block(R2, tau2, n, g, 2 * K, exact)
                    if g < gmax:
                        images[(g, 2 * K + 1)] = prepare_block(R2, tau2, n, g, 2 * K + 1, exact)
            # every reflector must be covered by exactly one block
            covered = set()
            for (g, k) in images:
                for t in range(QW):
                    if (g * QW + t, k) in refl:
                        covered.add((g * QW + t, k))
            missing = [key for key in refl if key not in covered and refl[key][2] != 0.0]
            assert not missing, missing[:5]
            Zt = Z0.copy()
            walk_wave(Zt, list(range(nrows)), R2, tau2, n, images, exact)
            err = np.abs(Zt - ref).max()
            print(f"n={n} exact={exact} blocks={len(images)} max err {err:.3e} (|Z| ~ {np.abs(ref).max():.2f})")




# ======================================================================================================================
# 32-row waves on v_mfma_f32_32x32x16_bf16 (csrc/q2slide.hip:qs32_*): lane (r = l & 31, h = l >> 5) keeps
# S[row r][8 q + 4 h .. + 3], q = 0..23; k order of a 16-deep step: k(h, j) = 16 step + 8 (j >> 2) + 4 h + (j & 3)
def kcol32(step, h, j):
    return 16 * step + 8 * (j >> 2) + 4 * h + (j & 3)


def mfma32(Af, Bf, C):
    """Af, Bf: [64][8]; C: [64][16].  A[row l&31][k = 8 (l>>5) + j], B[k][col l&31]; D[(e&3) + 8 (e>>2) + 4 (l>>5)][l&31]."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for ln in range(64):
        for j in range(8):
            A[ln & 31, 8 * (ln >> 5) + j] = Af[ln, j]
            B[8 * (ln >> 5) + j, ln & 31] = Bf[ln, j]
    D = A @ B
    out = C.copy()
    for ln in range(64):
        for e in range(16):
            out[ln, e] = np.float32(out[ln, e] + D[(e & 3) + 8 * (e >> 2) + 4 * (ln >> 5), ln & 31])
    return out


def mfma32_6(Ap, Bp, C):
    for a, b in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
        C = mfma32(Ap[a], Bp[b], C)
    return C


W2_LIST32 = [(ks, ta) for ks in range(8) for ta in range(2) if 16 * ks + 15 >= 32 * ta + 1]
U_LIST32 = [(wt, kt) for wt in range(4) for kt in range(4) if 16 * kt <= 32 * wt + 30 and 16 * kt + 15 >= 32 * wt - 64]
assert len(W2_LIST32) == 14 and len(U_LIST32) == 12, (len(W2_LIST32), len(U_LIST32))


def prepare_block32(R2, tau2, n, g, k, exact=False):
    Vw, taus = block_V(R2, tau2, n, g, k)
    S = (Vw.astype(np.float64) @ Vw.astype(np.float64).T).astype(np.float32)
    T = np.zeros((QW, QW), dtype=np.float32)
    for j in range(QW):
        T[j, j] = taus[j]
        for i in range(j - 1, -1, -1):
            T[i, j] = -taus[i] * np.float32(np.dot(S[i, i + 1:j + 1].astype(np.float64), T[i + 1:j + 1, j].astype(np.float64)))
    TV = (np.triu(T).astype(np.float64) @ Vw.astype(np.float64)).astype(np.float32)
    img = np.zeros((NFRAG, 64, 8), dtype=np.float32)
    f = 0
    for ks, ta in W2_LIST32:
        vals = np.array([[TV[32 * ta + (ln & 31), kcol32(ks, ln >> 5, j)] for j in range(8)] for ln in range(64)], dtype=np.float32)
        pcs = (vals, np.zeros_like(vals), np.zeros_like(vals)) if exact else split3(vals)
        for p in range(3):
            img[f + p] = pcs[p]
        f += 3
    for wt, kt in U_LIST32:
        vals = np.array([[Vw[kcol32(kt, ln >> 5, j), 32 * wt + (ln & 31)] for j in range(8)] for ln in range(64)], dtype=np.float32)
        pcs = (vals, np.zeros_like(vals), np.zeros_like(vals)) if exact else split3(vals)
        for p in range(3):
            img[f + p] = pcs[p]
        f += 3
    return img


def apply_block32(sw, Q0, img, exact=False):
    """sw: [24][64][4]; block on float4s Q0 .. Q0 + 15"""
    sp = (lambda v: (v, np.zeros_like(v), np.zeros_like(v))) if exact else split3
    acc2 = [np.zeros((64, 16), dtype=np.float32) for _ in range(2)]
    f = 0
    cur, Bp = -1, None
    for ks, ta in W2_LIST32:
        if ks != cur:
            Bp = sp(np.concatenate([sw[Q0 + 2 * ks], sw[Q0 + 2 * ks + 1]], axis=1))
            cur = ks
        acc2[ta] = mfma32_6((img[f], img[f + 1], img[f + 2]), Bp, acc2[ta])
        f += 3
    Wp = [sp(acc2[kt >> 1][:, 8 * (kt & 1):8 * (kt & 1) + 8]) for kt in range(4)]
    u, cur = None, -1
    for idx, (wt, kt) in enumerate(U_LIST32):
        if wt != cur:
            u = np.zeros((64, 16), dtype=np.float32)
            cur = wt
        u = mfma32_6((img[f], img[f + 1], img[f + 2]), Wp[kt], u)
        f += 3
        if idx + 1 == len(U_LIST32) or U_LIST32[idx + 1][0] != wt:
            for c in range(4):
                sw[Q0 + 4 * wt + c] = (sw[Q0 + 4 * wt + c] - u[:, 4 * c:4 * c + 4]).astype(np.float32)
    assert f == NFRAG


def load_unit32(Zt, rows, u, n):
    out = np.zeros((8, 64, 4), dtype=np.float32)
    for c in range(8):
        for ln in range(64):
            col = 64 * u + 8 * c + 4 * (ln >> 5)
            r = rows[ln & 31]
            if r is not None and col < n:
                out[c, ln] = Zt[r, col:col + 4]
    return out


def store_unit32(Zt, rows, u, n, regs):
    for c in range(8):
        for ln in range(64):
            col = 64 * u + 8 * c + 4 * (ln >> 5)
            r = rows[ln & 31]
            if r is not None and col < n:
                Zt[r, col:col + 4] = regs[c, ln]


def walk_wave32(Zt, rows, R2, tau2, n, images, exact):
    for K, gmax in passes(n):
        g = gmax
        sw = np.zeros((24, 64, 4), dtype=np.float32)
        sw[0:8] = load_unit32(Zt, rows, g + 2 * K, n)      # units G0 + 1, G0 + 2 lie outside the matrix
        while g >= 0:
            up = g - 1 + 2 * K
            nxt = load_unit32(Zt, rows, max(up, 0), n)
            apply_block32(sw, 0, images[(g, 2 * K)], exact)
            if g < gmax:
                apply_block32(sw, 8, images[(g, 2 * K + 1)], exact)
            store_unit32(Zt, rows, g + 2 * K + 2, n, sw[16:24])
            sw[16:24] = sw[8:16]
            sw[8:16] = sw[0:8]
            sw[0:8] = nxt
            g -= 1
        store_unit32(Zt, rows, 2 * K, n, sw[8:16])
        store_unit32(Zt, rows, 2 * K + 1, n, sw[16:24])


def main32():
    rng = np.random.default_rng(1)
    for n in (132, 200, 324):
        M = rng.standard_normal((n, n)); M = (M + M.T) / 2
        band = np.triu(np.tril(M, NB), -NB)
        d, e, refl, off = sb2st(band, NB, "wavefront")
        R2 = np.zeros((n, n), dtype=np.float32)
        tau2 = np.zeros((n, n // NB + 2), dtype=np.float32)
        for (s, k), (c0, v, tau) in refl.items():
            R2[s, c0:c0 + len(v)] = v
            tau2[s, k] = tau
        refl32 = {key: (c0, R2[key[0], c0:c0 + len(v)].astype(np.float64), float(tau2[key[0], key[1]]))
                  for key, (c0, v, tau) in refl.items()}
        nrows = 27
        Z0 = rng.standard_normal((nrows, n)).astype(np.float32) / np.sqrt(n)
        ref = reference(Z0, refl32, n)
        for exact in (True, False):
            images = {}
            for K, gmax in passes(n):
                for g in range(gmax, -1, -1):
                    images[(g, 2 * K)] = prepare_block32(R2, tau2, n, g, 2 * K, exact)
                    if g < gmax:
                        images[(g, 2 * K + 1)] = prepare_block32(R2, tau2, n, g, 2 * K + 1, exact)
            Zt = Z0.copy()
            rows = [r if r < nrows else None for r in range(32)]
            walk_wave32(Zt, rows, R2, tau2, n, images, exact)
            print(f"32-row: n={n} exact={exact} max err {np.abs(Zt - ref).max():.3e}")


if __name__ == "__main__":
    if "16" in sys.argv[1:] or len(sys.argv) == 1:
        main()
    if "32" in sys.argv[1:] or len(sys.argv) == 1:
        main32()
