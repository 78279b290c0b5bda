"""GPU parity of the wave-resident wavelet tiles (mono blocks, the middle of every block that holds a
window of 1024 ... 4096 samples; blocks whose length is a multiple of the lanes' run are covered whole).  Against the
LDS tiles alone (diagnostic option bit 4096 switches the wave-resident tiles off): bit for bit, outputs and block
statistics.  Against the oracle: fp32 tolerance of the wavelet path (1e-4 max-abs, >= 90 dB SNR of the difference)."""
import numpy as np
import pytest
import torch

import adn_oracle as orc
import synth
from gpu_util import engine, dev, host, check32, check64

pytestmark = pytest.mark.gpu

# odd, just below / at / above one window of every run length, multiples of 16 / 32 / 64 (windows go round the block)
# and not, shorter than any window, the benchmark's 144 000
LENS = np.array([144000, 143999, 1023, 1024, 1025, 2047, 4096, 4097, 5000, 7681, 12345, 100001, 3, 64, 20000, 8192, 4160,
                 2048, 1040, 12352, 132300, 7168, 3584, 960], dtype=np.int64)


def _run(e, x, off, lens, ablate, **kw):
    e._check(e.lib.adn_set_option(e._h, 2, ablate))
    try:
        y = torch.full_like(x, 7.0)
        st = e.wavelet_denoise(x, e._i64(off), e._i64(lens), y, e._i64(off), n_ch=1, want_stats=True,
                               max_frames=int(lens.max()), **kw)
        torch.cuda.synchronize()
        return host(y), [host(t) for t in st]
    finally:
        e._check(e.lib.adn_set_option(e._h, 2, 0))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("wavelet,level,thr", [("db4", 2, "soft"), ("db4", 1, "hard"), ("db4", 3, "soft"), ("db4", 4, "soft"),
                                               ("db8", 1, "soft"), ("db8", 2, "hard"), ("db8", 3, "soft"), ("db8", 4, "soft"),
                                               ("db8", 8, "soft"), ("db4", 6, "hard")])
def test_wave_resident_tiles_equal_lds_tiles(wavelet, level, thr, dtype):
    """(binary64 db8 at four levels and beyond takes its deepest halos from two lanes away.)"""
    e = engine()
    dt = torch.float32 if dtype == "f32" else torch.float64
    ut = np.uint32 if dtype == "f32" else np.uint64
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    off = np.concatenate([[0], np.cumsum(LENS + 7)[:-1]]).astype(np.int64)     # block starts at odd element offsets too
    total = int(off[-1] + LENS[-1] + 7)
    x = torch.randn(total, dtype=dt, device="cuda:0", generator=gen) * 0.1
    # digital silence (0 / 0 in the soft threshold), a quantised quiet block (tied statistics), one NaN, one infinity
    for blk, kind in ((6, "zero"), (9, "zero"), (15, "quant"), (19, "quant"), (16, "nan"), (7, "inf")):
        o, n = int(off[blk]), int(LENS[blk])
        if kind == "zero":
            x[o:o + n] = 0.0
        elif kind == "quant":
            x[o:o + n] = torch.round(x[o:o + n] * 40.0) / 32768.0
        elif kind == "nan":
            x[o + n // 2] = float("nan")
        else:
            x[o + n // 3] = float("inf")
    y_w, st_w = _run(e, x, off, LENS, 0, wavelet=wavelet, level=level, thr_mode=thr)
    y_l, st_l = _run(e, x, off, LENS, 4096, wavelet=wavelet, level=level, thr_mode=thr)
    assert np.array_equal(y_w.view(ut), y_l.view(ut))
    for a, b in zip(st_w, st_l):
        assert np.array_equal(a, b, equal_nan=True)
    # and the LDS tiles with the statistics in two passes (finest band in memory, two selections on it)
    y_2, st_2 = _run(e, x, off, LENS, 4096 | 512, wavelet=wavelet, level=level, thr_mode=thr)
    assert np.array_equal(y_w.view(ut), y_2.view(ut))
    for a, b in zip(st_w, st_2):
        assert np.array_equal(a, b, equal_nan=True)
    # elements between the blocks are nobody's to write
    gaps = np.ones(total, dtype=bool)
    for o, n in zip(off, LENS):
        gaps[o:o + n + (n & 1)] = False
    assert np.all(y_w[gaps] == 7.0)


@pytest.mark.timeout(300)
def test_wave_resident_tiles_binary64_against_oracle():
    """db4 level 2 (the reference's default transform) in the reference's precision: ring and end-tile block lengths."""
    e = engine()
    lens = np.array([12352, 9001, 8192, 20000], dtype=np.int64)
    off = (np.concatenate([[0], np.cumsum(lens)[:-1]]) + 3) // 4 * 4
    total = int(off[-1] + lens[-1] + 4)
    x64 = np.zeros(total)
    for i, (o, n) in enumerate(zip(off, lens)):
        x64[o:o + n] = synth.noisy_speech(int(n), 48000, 42, 4200 + i)
    y, (sigma, thresh, zeroed) = _run(e, dev(x64, torch.float64), off, lens, 0, wavelet="db4", level=2, thr_mode="soft")
    for i, (o, n) in enumerate(zip(off, lens)):
        want, s_or, t_or = orc.denoise_block(x64[o:o + n], "db4", 2, "soft")
        assert abs(sigma[i] - s_or) <= 1e-12 and abs(thresh[i] - t_or) <= 1e-12
        check64(y[o:o + n], want[:n])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("wavelet,level,thr", [("db4", 2, "soft"), ("db8", 4, "soft"), ("db8", 8, "soft"), ("db4", 3, "hard")])
def test_wave_resident_tiles_against_oracle(wavelet, level, thr):
    """Aligned blocks (16-byte aligned starts: the staged, coalesced stores) of lengths that take the ring of windows
    (12 352 = 64 x 193, 8192) and the windows-plus-end-tiles route (9001, 20 000 at db8 level 4: a multiple of 32,
    not of 64)."""
    e = engine()
    lens = np.array([12352, 9001, 8192, 20000], dtype=np.int64)
    off = (np.concatenate([[0], np.cumsum(lens)[:-1]]) + 3) // 4 * 4
    total = int(off[-1] + lens[-1] + 4)
    x64 = np.zeros(total)
    for i, (o, n) in enumerate(zip(off, lens)):
        x64[o:o + n] = synth.noisy_speech(int(n), 48000, 42, 4100 + i)
    x32 = x64.astype(np.float32)
    y, (sigma, thresh, zeroed) = _run(e, dev(x32, torch.float32), off, lens, 0, wavelet=wavelet, level=level, thr_mode=thr)
    for i, (o, n) in enumerate(zip(off, lens)):
        want, s_or, t_or, bands = orc.denoise_block(x32[o:o + n].astype(np.float64), wavelet, level, thr, return_bands=True)
        assert abs(sigma[i] - s_or) <= 1e-6 * max(1.0, abs(s_or))
        got = y[o:o + n]
        if thr == "soft":
            check32(got, want[:n])
        else:
            # a coefficient on the other side of the threshold in binary32 moves a few samples by about the threshold
            z_or = int(sum(np.sum(np.abs(b) < t_or) for b in bands[1:]))
            assert abs(int(zeroed[i]) - z_or) <= 4
            assert np.mean(np.abs(got - want[:n]) > 1e-4) <= 1e-3


@pytest.mark.timeout(300)
def test_wave_resident_tiles_many_blocks():
    """More blocks than one grid dimension holds (40 000 > 32 768: the block index comes from two grid dimensions)."""
    e = engine()
    nb, n = 40000, 1040                                   # 65 runs of 16 samples: the windows go round the block
    gen = torch.Generator(device="cuda:0").manual_seed(11)
    x = torch.randn(nb * n, dtype=torch.float32, device="cuda:0", generator=gen) * 0.1
    off = np.arange(nb, dtype=np.int64) * n
    lens = np.full(nb, n, dtype=np.int64)
    y_w, st_w = _run(e, x, off, lens, 0, wavelet="db4", level=2, thr_mode="soft")
    y_l, st_l = _run(e, x, off, lens, 4096, wavelet="db4", level=2, thr_mode="soft")
    assert np.array_equal(y_w.view(np.uint32), y_l.view(np.uint32))
    for a, b in zip(st_w, st_l):
        assert np.array_equal(a, b)
    xh = host(x)
    for b in (0, 32767, 32768, nb - 1):
        want, s_or, _ = orc.denoise_block(xh[b * n:(b + 1) * n].astype(np.float64), "db4", 2, "soft")
        assert abs(st_w[0][b] - s_or) <= 1e-6 * max(1.0, abs(s_or))
        check32(y_w[b * n:(b + 1) * n], want)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("wavelet,level", [("db4", 2), ("db8", 4), ("db8", 8), ("db4", 5)])
def test_wave_resident_tiles_random_lengths(wavelet, level, dtype):
    """400 blocks of random lengths in one batch (a third of them multiples of 16, 32 or 64, the lengths whose windows
    go round the block; block starts at any element offset): samples and block statistics equal those of the LDS
    tiles bit for bit."""
    e = engine()
    dt = torch.float32 if dtype == "f32" else torch.float64
    ut = np.uint32 if dtype == "f32" else np.uint64
    rng = np.random.default_rng(20 + level)
    lens = rng.integers(1, 30000, size=400).astype(np.int64)
    mult = rng.choice([16, 32, 64], size=400)
    pick = rng.random(400) < 1.0 / 3.0
    lens[pick] = np.maximum(mult[pick], lens[pick] // mult[pick] * mult[pick])
    gap = rng.integers(0, 9, size=400)
    off = np.concatenate([[0], np.cumsum(lens + (lens & 1) + gap)[:-1]]).astype(np.int64)
    total = int(off[-1] + lens[-1] + 9)
    gen = torch.Generator(device="cuda:0").manual_seed(31 + level)
    x = torch.randn(total, dtype=dt, device="cuda:0", generator=gen) * 0.1
    y_w, st_w = _run(e, x, off, lens, 0, wavelet=wavelet, level=level, thr_mode="soft")
    y_l, st_l = _run(e, x, off, lens, 4096, wavelet=wavelet, level=level, thr_mode="soft")
    bad = [int(n) for o, n in zip(off, lens) if not np.array_equal(y_w[o:o + n].view(ut), y_l[o:o + n].view(ut))]
    assert not bad, bad[:20]
    assert np.array_equal(y_w.view(ut), y_l.view(ut))
    for a, b in zip(st_w, st_l):
        assert np.array_equal(a, b, equal_nan=True)
