"""Index model of the fused DWT tile kernels (TEST INFRASTRUCTURE, numpy).

The HIP tile kernels of csrc/adn_dwt_tiles.hpp keep, for every in-tile level, a
CANONICAL CONTIGUOUS RANGE (start s, count c, indices taken modulo the level length)
of the approximation / detail arrays in LDS.  This model executes exactly the same
plan and index arithmetic on numpy arrays so that the range planning, the interior
(direct indexing) path and the boundary (exact modular indexing) path can be checked
against the oracle for any block length before / beside the GPU tests.
"""
import numpy as np

import adn_oracle as orc


def level_lengths(n, levels):
    """N_l (band length) and M_l (even-extended length) for l = 0..levels."""
    N, M = [n], []
    for _ in range(levels):
        M.append(N[-1] + (N[-1] & 1))
        N.append(M[-1] // 2)
    M.append(N[-1] + (N[-1] & 1))
    return N, M


class Rng:
    """Canonical contiguous range: indices (s + i) mod N, i < c."""

    def __init__(self, s, c, N):
        if c >= N:
            s, c = 0, N
        self.s, self.c, self.N = s % N, c, N

    def local(self, q):
        li = q - self.s
        if li < 0:
            li += self.N
        assert 0 <= li < self.c, ("range miss", q, self.s, self.c, self.N)
        return li

    def wraps(self):
        return self.s + self.c > self.N

    def canon(self, i):
        q = self.s + i
        return q - self.N if q >= self.N else q


def inv_source_range(out_rng_lo, out_rng_cnt, F, N_src):
    """Coefficients (level l) needed to synthesise outputs [lo, lo+cnt) of level l-1 (unwrapped lo)."""
    klo, khi = inv_source_unwrapped(out_rng_lo, out_rng_cnt, F)
    return Rng(klo, khi - klo + 1, N_src)


def inv_source_unwrapped(lo, cnt, F):
    """Unwrapped first / last coefficient index feeding outputs [lo, lo+cnt).  One extra index
    on each side: when the output range wraps around an ODD length the coefficient index
    of the wrapped outputs shifts by one (2 * N_l = N_{l-1} + 1)."""
    return (lo - F // 2) // 2 - 1, (lo + cnt - 1 + F // 2 - 1) // 2 + 1


def fwd_source_range(g, F, N_src, M_src):
    """Samples of level l-1 needed to analyse coefficients g (canonical range at level l)."""
    lo = 2 * g.s - F // 2 + 1
    cnt = 2 * (g.c - 1) + F + 2            # +2: the replicated pad sample shifts indices by one
    lo_c = lo % M_src
    if lo_c == N_src:                      # landed on the pad position
        lo_c = N_src - 1
    return Rng(lo_c - 1, cnt + 1, N_src)


def plan_tile(n, n0, n1, F, TL, mode):
    """Ranges of one tile.  mode 'apply': outputs y[n0:n1); mode 'stats': owned coefficients."""
    N, M = level_lengths(n, TL)
    I = [None] * (TL + 1)                  # inverse-needed (or owned) ranges per level
    if mode == "apply":
        lo, cnt = n0, n1 - n0
        for l in range(1, TL + 1):
            I[l] = inv_source_range(lo, cnt, F, N[l])
            # next level works on the unwrapped start of this range
            klo, khi = inv_source_unwrapped(lo, cnt, F)
            lo, cnt = klo, khi - klo + 1
    else:
        lo, hi = n0, n1
        for l in range(1, TL + 1):
            lo, hi = lo // 2, (hi + 1) // 2
            I[l] = Rng(lo, min(hi, N[l]) - lo, N[l])
    G = [None] * (TL + 1)                  # ranges the forward pass computes (A and D) per level
    G[TL] = I[TL]
    for l in range(TL - 1, -1, -1):
        G[l] = fwd_source_range(G[l + 1], F, N[l], M[l])
    return N, M, I, G


def run_tile(x, n0, n1, wavelet, TL, thresh, thr_mode, top=None, mode="apply"):
    """One tile of the fused kernel.  Returns y[n0:n1) (apply) or dict of owned coefficients (stats).
    top: optional replacement for the level-TL approximation (output of the deep kernel)."""
    h, g, rl, rh = orc.filters(wavelet)
    F = h.size
    n = x.shape[0]
    N, M, I, G = plan_tile(n, n0, n1, F, TL, mode)
    interior = all(not G[l].wraps() for l in range(TL + 1)) and all(
        2 * G[l + 1].s - F // 2 + 1 >= 0 and 2 * (G[l + 1].s + G[l + 1].c - 1) + F // 2 <= N[l] - 1
        for l in range(TL))
    if mode == "apply":
        lo, cnt = n0, n1 - n0
        for l in range(1, TL + 1):
            klo, khi = inv_source_unwrapped(lo, cnt, F)
            interior = interior and klo >= 0 and khi <= N[l] - 1
            lo, cnt = klo, khi - klo + 1
    # ---- level 0 from memory
    A = [None] * (TL + 1)
    D = [None] * (TL + 1)
    A[0] = np.array([x[G[0].canon(i)] for i in range(G[0].c)])
    # ---- forward
    for l in range(1, TL + 1):
        src, dst = G[l - 1], G[l]
        A[l] = np.zeros(dst.c)
        D[l] = np.zeros(dst.c)
        for i in range(dst.c):
            k = dst.canon(i)
            sa = sd = 0.0
            if interior:
                off = 2 * dst.s + F // 2 - src.s
                for j in range(F):
                    v = A[l - 1][2 * i + off - j]
                    sa += h[j] * v
                    sd += g[j] * v
            else:
                idx = (2 * k + F // 2) % M[l - 1]
                for j in range(F):
                    q = min(idx, N[l - 1] - 1)
                    v = A[l - 1][src.local(q)]
                    sa += h[j] * v
                    sd += g[j] * v
                    idx = M[l - 1] - 1 if idx == 0 else idx - 1
            A[l][i], D[l][i] = sa, sd
    if mode == "stats":
        out = {}
        for l in range(1, TL + 1):
            own = I[l]
            out["d%d" % l] = np.array([D[l][G[l].local(own.canon(i))] for i in range(own.c)])
            out["a%d" % l] = np.array([A[l][G[l].local(own.canon(i))] for i in range(own.c)])
            out["own%d" % l] = (own.s, own.c)
        return out
    # ---- shrink + inverse.  R[l] holds the reconstruction over I[l] (R[TL] = A[TL] or `top`)
    R = [None] * (TL + 1)
    if top is None:
        R[TL] = np.array([A[TL][G[TL].local(I[TL].canon(i))] for i in range(I[TL].c)])
    else:
        R[TL] = np.array([top[I[TL].canon(i)] for i in range(I[TL].c)])
    for l in range(TL, 0, -1):
        dst_rng = I[l - 1] if l > 1 else Rng(n0, n1 - n0, 2 * N[1])
        out = np.zeros(dst_rng.c)
        K = N[l]
        for i in range(dst_rng.c):
            nn = dst_rng.canon(i)
            par = (nn + F // 2 - 1) & 1
            k = ((nn + F // 2 - 1 - par) >> 1) % K
            s = 0.0
            for j in range(par, F, 2):
                dv = D[l][G[l].local(k)]
                dv = float(orc.threshold(np.array([dv]), thresh, thr_mode)[0])
                s += rl[j] * R[l][I[l].local(k)] + rh[j] * dv
                k = K - 1 if k == 0 else k - 1
            out[i] = s
        R[l - 1] = out
    return R[0]


def denoise_block_tiled(x, wavelet="db4", level=2, thr_mode="soft", tile=64, TL=None):
    """Whole block through the tile model (levels above TL are done on whole arrays like the
    deep kernel).  Returns (y, sigma, thresh)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    TL = min(level, 3) if TL is None else TL
    N, M = level_lengths(n, TL)
    # stats pass: owned coefficients of every tile
    d1 = np.zeros(N[1])
    aT = np.zeros(N[TL])
    n_out = M[0]
    for t0 in range(0, n, tile):
        t1 = min(t0 + tile, n)
        st = run_tile(x, t0, t1, wavelet, TL, 0.0, thr_mode, mode="stats")
        s, c = st["own1"]
        d1[s:s + c] = st["d1"]
        s, c = st["own%d" % TL]
        aT[s:s + c] = st["a%d" % TL]
    sigma = orc.mad(d1)
    with np.errstate(divide="ignore"):
        thresh = sigma * np.sqrt(2 * np.log(n))
    top = None
    if level > TL:
        co = orc.wavedec(aT, wavelet, level - TL, "per")
        co = [co[0]] + [orc.threshold(c, thresh, thr_mode) for c in co[1:]]
        top = orc.waverec(co, wavelet, "per")[:N[TL]]
    y = np.zeros(n_out)
    for t0 in range(0, n_out, tile):
        t1 = min(t0 + tile, n_out)
        y[t0:t1] = run_tile(x, t0, t1, wavelet, TL, thresh, thr_mode, top=top, mode="apply")
    return y, sigma, thresh
