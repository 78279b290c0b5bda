"""Index model of the one-pass threshold statistics of mono blocks (TEST INFRASTRUCTURE, numpy).

csrc/adn_dwt_mad1p.hpp forms the finest detail band of a mono block on the fly
(denoise.py:77-83: the threshold needs only cD1 of pywt.wavedec(block, mode='per')):
  * SrcMono::at - one coefficient with the periodization index,
        d[k] = sum_j g[j] * xe[(2 k + F/2 - j) mod M],  M = n rounded up to even,
        xe = the block with its last sample repeated when n is odd;
  * k_d1_gather_direct - tiles of TILE coefficients; in a tile that neither wraps nor
    touches the repeated sample and whose first sample is 16-byte aligned, thread tid
    forms runs of ADJ neighbouring coefficients from aligned pieces of VW samples:
        run rr = q * NT + tid, coefficients k0 + ADJ rr + a, samples base + 2 ADJ rr + i,
        i < F + 2 ADJ, base = 2 k0 - F/2, tap j of coefficient a meets sample i = F + 2 a - j.
This model executes the same index arithmetic so that it can be checked against the oracle's
wavedec for any block length, filter length and buffer alignment without a GPU.
"""
import numpy as np

import adn_oracle as orc

TILE, NT, BATCH = 4096, 256, 16


def at(x, g, k):
    """SrcMono::at: coefficient k of the finest detail band of block x (taps g)."""
    n, F = len(x), len(g)
    M = n + (n & 1)
    idx0 = 2 * k + F // 2
    while idx0 >= M:
        idx0 -= M
    s = 0.0
    for j in range(F):
        idx = idx0 - j
        while idx < 0:
            idx += M
        s += g[j] * x[idx if idx < n else n - 1]
    return s


def gather_all(buf, off, n, g, adj=4, vw=4):
    """All coefficients of the block buf[off : off + n] in the order and through the index paths of
    k_d1_gather_direct; returns (d, n_direct_tiles, n_edge_tiles).  vw: samples per 16-byte piece."""
    F = len(g)
    x = buf[off:off + n]
    nc = (n + 1) // 2
    d = np.full(nc, np.nan)
    direct_tiles = edge_tiles = 0
    for k0 in range(0, nc, TILE):
        base = 2 * k0 - F // 2
        direct = base >= 0 and base + 2 * TILE + F <= n and (off + base) % vw == 0
        if direct:
            direct_tiles += 1
            nch = (F + 2 * adj) // vw
            for q in range(BATCH // adj):
                for tid in range(NT):
                    rr = q * NT + tid
                    first = base + 2 * adj * rr
                    assert (off + first) % vw == 0               # every piece is aligned
                    v = x[first:first + nch * vw]
                    assert len(v) == nch * vw                    # and inside the block
                    for a in range(adj):
                        k = k0 + adj * rr + a
                        s = 0.0
                        for j in range(F):
                            s += g[j] * v[F + 2 * a - j]
                        assert np.isnan(d[k])                    # every coefficient once
                        d[k] = s
        else:
            edge_tiles += 1
            for u in range(BATCH):
                for tid in range(NT):
                    k = k0 + adj * ((u // adj) * NT + tid) + u % adj
                    if k < nc:
                        assert np.isnan(d[k])
                        d[k] = at(x, g, k)
    return d, direct_tiles, edge_tiles


def reference_detail(x, wavelet):
    """cD1 of the oracle's periodized wavedec."""
    return orc.wavedec(np.asarray(x, dtype=np.float64), wavelet, 1, "per")[-1]
