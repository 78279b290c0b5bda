"""GPU parity of the streaming select of long power rows (Wiener stage, files of more than about four minutes: every
16th group of four frames brackets the order statistic, one pass over the rows gathers the candidates inside the
bracket, the exact select runs on them; rows whose bracket misses are left to the register-resident select).
Diagnostic option bit 16384 sends rows of any length this way, bit 32768 none; bit 65536 gives the rows the brackets
miss to the radix selection that serves rows beyond the register-resident select's reach (90 112 frames in binary32,
49 152 in binary64) instead of the register-resident select.  The order statistics are the same
keys on either route (gain rows bit for bit when no voice-activity mean is involved); the voice-activity mean adds
its terms in another order (binary32: 2e-6, binary64: 1e-12)."""
import numpy as np
import pytest
import torch

import adn_oracle as orc
import synth
from gpu_util import engine, dev, host, check32, check64

pytestmark = pytest.mark.gpu
SR = 48000


def _wiener(e, bits, *a, **kw):
    e._check(e.lib.adn_set_option(e._h, 2, bits))
    try:
        e.kernel_timing(True)
        y, l, g = e.wiener(*a, want_gain=True, **kw)
        torch.cuda.synchronize()
        names = list(e.kernel_times())
        return y, g, names
    finally:
        e.kernel_timing(False)
        e._check(e.lib.adn_set_option(e._h, 2, 0))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("n_ch", [1, 2])
def test_streaming_row_select_equals_register_select(dt, n_ch):
    """A ragged batch through both routes: speech in noise (30 s), a pure tone (most rows hold a few distinct tiny
    values), digital silence (every key equal: the whole row is its own candidate list), noise quantised to 16-bit
    steps (ties), a signal with one NaN sample (NaN rows: left to the register-resident select), a signal one window
    long (one frame: no sample to take a bracket from), one of 3 s, and noise in bursts that only the sampled frames see."""
    e = engine()
    npdt = np.float64 if dt == torch.float64 else np.float32
    rng = np.random.default_rng(77)
    lens = [30 * SR + 123, 20 * SR, 12 * SR + 5, 25 * SR, 9 * SR, 2048, 3 * SR, 40 * SR]
    t = np.arange(lens[1]) / SR
    sigs = [synth.noisy_speech(lens[0], SR, 42, 501),
            0.5 * np.sin(2 * np.pi * 1000.0 * t),
            np.zeros(lens[2]),
            np.round(rng.standard_normal(lens[3]) * 40.0) / 32768.0,
            synth.noisy_speech(lens[4], SR, 42, 502),
            rng.standard_normal(lens[5]) * 0.1,
            synth.noisy_speech(lens[6], SR, 42, 503),
            rng.standard_normal(lens[7]) * 0.1]
    # bursts at the period of the sample (every 16th group of four frames = 64 hops): the sampled frames are all loud,
    # nine tenths of the others quiet, so that the brackets miss and the rows go back to the register-resident select
    pos = np.arange(lens[7]) % (64 * 512)
    sigs[7] *= np.where(pos < 2048 + 3 * 512, 1.0, 0.01)
    sigs[4][4 * SR + 17] = np.nan
    if n_ch == 2:
        sigs = [np.stack([s, 0.7 * s + 0.01 * rng.standard_normal(len(s)) * (np.abs(s).max() > 0)], axis=1).reshape(-1) for s in sigs]
    xs = dev(np.concatenate(sigs).astype(npdt), dt)
    for vad in (True, False):
        y0, g0, k0 = _wiener(e, 32768, xs, lens, SR, n_ch=n_ch, vad_guided=vad)
        assert not any("k_rowq" in k for k in k0)
        g0h, y0h = host(g0), host(y0)
        for bits, last in ((16384, "waves per row"), (16384 | 65536, "k_rowq_exact")):
            y1, g1, k1 = _wiener(e, bits, xs, lens, SR, n_ch=n_ch, vad_guided=vad)
            assert any("k_rowq_classify" in k for k in k1) and any(last in k for k in k1)
            g1h, y1h = host(g1), host(y1)
            assert np.array_equal(np.isnan(g0h), np.isnan(g1h)) and np.array_equal(np.isnan(y0h), np.isnan(y1h))
            if vad:
                tol = 1e-12 if dt == torch.float64 else 2e-6
                assert np.nanmax(np.abs(g0h - g1h)) <= tol
                assert np.nanmax(np.abs(y0h - y1h)) <= tol
            else:
                assert np.array_equal(g0h, g1h, equal_nan=True)
                assert np.array_equal(y0h, y1h, equal_nan=True)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dt,seconds", [(torch.float32, 600), (torch.float64, 300)])
def test_streaming_row_select_long_file_against_oracle(dt, seconds):
    """One mono file of 600 s at 48 kHz in binary32 (56 247 frames per row) and one of 300 s in binary64 (28 122
    frames): lengths that take the streaming route by themselves.  Through the Wiener stage, with and without
    voice-activity guidance: against the oracle and against the register-resident select of the same rows."""
    e = engine()
    f64 = dt == torch.float64
    n = seconds * SR
    x = np.concatenate([synth.noisy_speech(60 * SR, SR, 42, 600 + i) * (0.5 + 0.1 * i) for i in range(seconds // 60)])
    x = x.astype(np.float64 if f64 else np.float32)
    xd = dev(x, dt)
    tol = 1e-12 if f64 else 2e-6
    for vad in (True, False):
        y1, g1, k1 = _wiener(e, 0, xd, [n], SR, vad_guided=vad)
        assert any("k_rowq_classify" in k for k in k1)
        y0, g0, k0 = _wiener(e, 32768, xd, [n], SR, vad_guided=vad)
        assert not any("k_rowq_classify" in k for k in k0)
        if vad:
            assert float((g0 - g1).abs().max()) <= tol and float((y0 - y1).abs().max()) <= tol
        else:
            assert torch.equal(g0, g1) and torch.equal(y0, y1)
        want, gain_or = orc.wiener(x.astype(np.float64), SR, vad_guided=vad)
        if f64:
            check64(host(y1), want)
            check64(host(g1)[0], gain_or, 1e-12)
        else:
            check32(host(y1), want)
            assert np.max(np.abs(host(g1)[0] - gain_or)) <= 2e-5


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dt,seconds", [(torch.float32, 1200), (torch.float64, 660)])
def test_rows_beyond_the_register_select_against_oracle(dt, seconds):
    """One mono file of 20 minutes in binary32 (112 497 frames per row) and one of 11 minutes in binary64 (61 872
    frames): rows no register-resident select holds.  They stay in the grouped layout, selected by the streaming
    route with the radix selection behind it.  Against the oracle, and against the transposed rows with the
    long-segment select (diagnostic option bit 64; the route of such files before, and of files beyond 786 432 /
    393 216 frames still)."""
    e = engine()
    f64 = dt == torch.float64
    n = seconds * SR
    x = np.concatenate([synth.noisy_speech(60 * SR, SR, 42, 700 + i) * (0.4 + 0.05 * (i % 9)) for i in range(seconds // 60)])
    x = x.astype(np.float64 if f64 else np.float32)
    xd = dev(x, dt)
    tol = 1e-12 if f64 else 2e-6
    for vad in (True, False):
        y1, g1, k1 = _wiener(e, 0, xd, [n], SR, vad_guided=vad)
        assert any("k_rowq_classify" in k for k in k1) and any("k_rowq_exact" in k for k in k1)
        assert not any("k_rows_fast" in k or "k_stft_rows" in k or "k_big" in k for k in k1)
        y0, g0, k0 = _wiener(e, 64, xd, [n], SR, vad_guided=vad)
        assert not any("k_rowq" in k for k in k0)
        assert float((g0 - g1).abs().max()) <= tol and float((y0 - y1).abs().max()) <= tol
        if not vad and not f64:
            assert torch.equal(g0, g1)                      # the same keys on both routes
        want, gain_or = orc.wiener(x.astype(np.float64), SR, vad_guided=vad)
        if f64:
            check64(host(y1), want)
            check64(host(g1)[0], gain_or, 1e-12)
        else:
            check32(host(y1), want)
            assert np.max(np.abs(host(g1)[0] - gain_or)) <= 2e-5


@pytest.mark.timeout(600)
@pytest.mark.parametrize("seconds", [3000, 6000])
def test_rows_of_hours_equal_transposed_route(seconds):
    """One mono binary32 file of 50 minutes (281 247 frames per row: candidate lists of 16 384) and one of 100 minutes
    (562 497 frames: lists of 32 768) through the Wiener stage on the grouped rows with the streaming select, against
    the transposed rows with the long-segment select (diagnostic option bit 64), which the oracle tests pin at
    shorter lengths: the same order statistics (gain rows bit for bit without voice-activity guidance)."""
    e = engine()
    n = seconds * SR
    gen = torch.Generator(device="cuda:0").manual_seed(seconds)
    t = torch.arange(n, device="cuda:0", dtype=torch.float32) / SR
    x = torch.randn(n, device="cuda:0", dtype=torch.float32, generator=gen) * (0.02 + 0.3 * torch.sin(2 * np.pi * t / 97.0).abs())
    del t
    for vad in (True, False):
        y1, g1, k1 = _wiener(e, 0, x, [n], SR, vad_guided=vad)
        assert any("k_rowq_classify" in k for k in k1) and not any("k_rows_fast" in k or "k_big" in k for k in k1)
        y0, g0, k0 = _wiener(e, 64, x, [n], SR, vad_guided=vad)
        assert not any("k_rowq" in k for k in k0)
        if vad:
            assert float((g0 - g1).abs().max()) <= 2e-6 and float((y0 - y1).abs().max()) <= 2e-6
        else:
            assert torch.equal(g0, g1) and torch.equal(y0, y1)
