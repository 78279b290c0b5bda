"""CPU check of the numpy index model of the fused DWT tile kernels (oracle/dwt_tile_model.py)
against the oracle: canonical ranges, interior and boundary index paths, deep levels.  The HIP
kernels implement exactly this plan; the GPU parity tests cover them on the device."""
import numpy as np
import pytest

import adn_oracle as orc
import dwt_tile_model as model


@pytest.mark.parametrize("wavelet,level,n,tile,TL", [
    ("db4", 2, 333, 64, 2), ("db4", 2, 512, 128, 2), ("db4", 3, 37, 64, 3), ("db4", 2, 1, 64, 2),
    ("db4", 2, 2, 64, 2), ("db8", 3, 257, 64, 3), ("db8", 5, 1001, 128, 3), ("db8", 4, 1000, 256, 4),
    ("db8", 8, 2003, 512, 4), ("db8", 2, 17, 64, 2), ("db8", 3, 9, 8, 3)])
def test_tile_model_matches_oracle(wavelet, level, n, tile, TL):
    rng = np.random.default_rng(n * 31 + level)
    x = rng.standard_normal(n)
    for mode in ("soft", "hard"):
        y, sigma, thresh = model.denoise_block_tiled(x, wavelet, level, mode, tile, TL)
        want, s_or, t_or = orc.denoise_block(x, wavelet, level, mode)
        assert sigma == s_or and thresh == t_or
        assert y.shape == want.shape
        assert np.max(np.abs(y - want)) <= 1e-12


def test_interior_tiles_exist_and_plan_is_in_bounds():
    """A long block has interior tiles (direct indexing) as well as wrapped ones, and every
    planned range fits the compile-time capacities the kernel derives from (S, F, TL)."""
    n, F, TL, S = 40001, 16, 4, 4096
    ic = [S]
    for l in range(1, TL + 1):
        ic.append((ic[-1] + F) // 2 + 3)
    gc = [0] * (TL + 1)
    gc[TL] = ic[TL]
    for l in range(TL - 1, -1, -1):
        gc[l] = 2 * (gc[l + 1] - 1) + F + 3
    kinds = set()
    for t0 in range(0, n + 1, S):
        t1 = min(t0 + S, n + 1)
        N, M, I, G = model.plan_tile(n, t0, t1, F, TL, "apply")
        for l in range(TL + 1):
            assert G[l].c <= gc[l]
            if l:
                assert I[l].c <= ic[l]
        kinds.add(any(G[l].wraps() for l in range(TL + 1)))
    assert kinds == {True, False}


@pytest.mark.parametrize("wavelet,n,off,vw", [
    ("db4", 30000, 0, 4), ("db4", 30001, 4, 4), ("db4", 24577, 3, 4), ("db8", 30000, 8, 4), ("db8", 16401, 0, 4),
    ("db4", 30000, 2, 2), ("db8", 30000, 1, 2), ("db4", 9, 0, 4), ("db4", 1, 0, 4), ("db8", 2, 5, 4), ("db4", 8200, 0, 4)])
def test_one_pass_statistics_gather_model(wavelet, n, off, vw):
    """Index model of the gather pass of the one-pass threshold statistics (oracle/d1_gather_model.py): every
    finest-band coefficient of a mono block is formed exactly once, through the aligned-piece path in interior
    tiles and the periodization index elsewhere, and equals the oracle's cD1 - for odd and tiny blocks, both
    filter lengths, aligned and unaligned block offsets, four and two samples per 16-byte piece."""
    import d1_gather_model as gm
    rng = np.random.default_rng(n + 7 * off)
    buf = rng.standard_normal(off + n + 8)
    g = orc.filters(wavelet)[1]
    d, n_direct, n_edge = gm.gather_all(buf, off, n, g, adj=4, vw=vw)
    want = gm.reference_detail(buf[off:off + n], wavelet)
    assert d.shape == want.shape and not np.any(np.isnan(d))
    assert np.max(np.abs(d - want)) <= 1e-12
    if n >= 3 * 8192 and off % vw == 0:
        assert n_direct >= 1 and n_edge >= 2            # first and last tile wrap, the middle ones are direct
    if off % vw:
        assert n_direct == 0
