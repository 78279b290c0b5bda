"""One recording processed in K time chunks (the split a recording too long for one device needs) against the
one-call result on the whole recording and against the oracle on windows round the seams, through the C ABI.
All chunks live on the one GPU and are walked in lock step (LocalReducer): the same code path as one chunk per
rank, with the all-reduce replaced by a sum over the list."""
import numpy as np
import pytest
import torch

import adn_oracle as orc
import synth
from gpu_util import engine, dev, host, check64, check32
from adn_hip import chunked as ck

pytestmark = pytest.mark.gpu
SR = 16000


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("wavelet,level,taps,n_chunks", [("db4", 2, 8, 3), ("db8", 8, 16, 4)])
def test_wavelet_block_in_chunks(dtype, wavelet, level, taps, n_chunks):
    e = engine()
    dt = torch.float64 if dtype == "f64" else torch.float32
    n = 160000                                            # one block; a multiple of 2^8
    x = dev(synth.noisy_speech(n, SR, 42, 900), dt)
    out = torch.empty_like(x)
    sigma1, thresh1, _ = e.wavelet_denoise(x, e._i64([0]), e._i64([n]), out, e._i64([0]), wavelet=wavelet, level=level,
                                           want_stats=True, max_frames=n)
    owned, halo = ck.plan_wavelet_chunks(n, n_chunks, level, taps)
    bufs = [ck.wavelet_chunk_buffer(x, a, b, halo) for a, b in owned]
    lens = [b - a for a, b in owned]
    red = ck.LocalReducer()
    sigma, thresh = ck.wavelet_block_threshold(e, bufs, lens, halo, n, wavelet, red)
    assert float(sigma[0]) == float(sigma1[0]), (float(sigma[0]), float(sigma1[0]))
    assert abs(float(thresh[0]) - float(thresh1[0])) <= 4e-16 * float(thresh1[0])
    # the one-call threshold, so that the comparison below is of the transform alone
    parts = ck.wavelet_block_apply(e, bufs, lens, halo, thresh1.to(torch.float64), wavelet, level)
    y = torch.cat(parts)
    diff = float((y - out).abs().max())
    print("%s %s level %d in %d chunks (halo %d): max |chunked - one call| = %.3g" % (dtype, wavelet, level, len(owned), halo, diff))
    assert torch.equal(y, out)
    # oracle on windows round every seam
    want = orc.denoise_block(host(x).astype(np.float64), wavelet, level)[0]
    chk = check64 if dtype == "f64" else check32
    for a, _ in owned[1:]:
        chk(host(y[a - 4096:a + 4096]), want[a - 4096:a + 4096])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n_chunks", [2, 5])
def test_spectral_subtract_in_chunks(dtype, n_chunks):
    e = engine()
    dt = torch.float64 if dtype == "f64" else torch.float32
    n = 20 * SR + 333
    xh = synth.noisy_speech(n, SR, 42, 910)
    x = dev(xh, dt)
    prof = e.noise_psd(dev(synth.white_noise(2 * SR, 42, 911), dt), [2 * SR])[0]
    y1, l1 = e.spectral_subtract(x, [n], prof)
    hops, frames = ck.plan_stft_chunks(n, n_chunks)
    pieces = []
    for h0, h1 in hops:
        s0, s1, keep = ck.stft_chunk_window(h0, h1, frames, n)
        pieces.append(ck.spectral_subtract_chunk(e, x[s0:s1].contiguous(), keep, (h1 - h0) * 512, prof))
    y = torch.cat(pieces)
    assert y.numel() == int(l1[0])
    assert torch.equal(y, y1)
    want = orc.spectral_subtract(host(x).astype(np.float64), host(prof).astype(np.float64))
    chk = check64 if dtype == "f64" else check32
    for h0, _ in hops[1:]:
        s = h0 * 512
        chk(host(y[s - 4096:s + 4096]), want[s - 4096:s + 4096])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("vad_guided", [True, False])
def test_wiener_in_chunks(dtype, vad_guided):
    e = engine()
    dt = torch.float64 if dtype == "f64" else torch.float32
    n = 24 * SR + 77
    x = dev(synth.noisy_speech(n, SR, 42, 920), dt)
    y1, l1, g1 = e.wiener(x, [n], SR, want_gain=True, vad_guided=vad_guided)
    n_chunks = 4
    hops, frames = ck.plan_stft_chunks(n, n_chunks)
    fl = int(SR * 20 / 1000.0)
    stats = []
    for i, (h0, h1) in enumerate(hops):
        t0, t1 = h0, (frames if i == len(hops) - 1 else h1)       # frames owned for the statistics
        s0 = t0 * 512
        s1 = n if t1 == frames else (t1 - 1) * 512 + 2048
        vb = max((t0 * 512 // fl - 5) * fl, 0)                    # five VAD frames of hangover
        stats.append(ck.wiener_chunk_stats(e, x[s0:s1].contiguous(), t1 - t0, x[vb:s1].contiguous(), vb, n, t0, SR,
                                           vad_guided=vad_guided))
    gain = ck.wiener_gain_from_chunks(e, stats, frames, dt, ck.LocalReducer(), vad_guided=vad_guided)
    gdiff = float((gain - g1[0]).abs().max())
    print("%s vad=%s: max |gain chunked - one call| = %.3g" % (dtype, vad_guided, gdiff))
    if vad_guided:       # the masked mean is summed in another order (and in binary64) than the one-call kernel's
        assert gdiff <= (1e-12 if dtype == "f64" else 2e-6)
    elif dtype == "f64":
        # binary64: the chunks' rows come from the shape-generic transform (adn_stft_rows_f64), the one call's from the
        # register-resident one: two correctly rounded binary64 computations of the same spectra
        assert gdiff <= 1e-12
    else:
        assert torch.equal(gain, g1[0])
    pieces = []
    for h0, h1 in hops:
        s0, s1, keep = ck.stft_chunk_window(h0, h1, frames, n, reach=0)
        pieces.append(ck.wiener_apply_chunk(e, x[s0:s1].contiguous(), keep, (h1 - h0) * 512, g1[0]))
    y = torch.cat(pieces)
    assert y.numel() == int(l1[0])
    assert torch.equal(y, y1)                                     # same gain -> same samples, seams included
    # chunked end to end (own gain) against the oracle round the seams
    pieces = []
    for h0, h1 in hops:
        s0, s1, keep = ck.stft_chunk_window(h0, h1, frames, n, reach=0)
        pieces.append(ck.wiener_apply_chunk(e, x[s0:s1].contiguous(), keep, (h1 - h0) * 512, gain))
    y = torch.cat(pieces)
    if vad_guided:
        want, _ = orc.wiener(host(x).astype(np.float64), SR)
        chk = check64 if dtype == "f64" else check32
        for h0, _ in hops[1:]:
            s = h0 * 512
            chk(host(y[s - 4096:s + 4096]), want[s - 4096:s + 4096])


def test_order_statistics_over_ragged_parts():
    """The distributed selection alone: ragged segments scattered unevenly over 3 holders, ties, negative
    values, an empty part; every rank of a few segments against numpy's sort."""
    e = engine()
    rng = np.random.default_rng(5)
    for dt, npdt in ((torch.float32, np.float32), (torch.float64, np.float64)):
        seg_vals = [np.round(rng.standard_normal(m) * 3, 1).astype(npdt) for m in (1, 7, 1000, 4097)]
        parts = []
        for h in range(3):
            cuts = [(len(v) * h // 3, len(v) * (h + 1) // 3) if h < 2 else (len(v) * 2 // 3, len(v)) for v in seg_vals]
            chunk = [v[a:b] for v, (a, b) in zip(seg_vals, cuts)]
            lens = np.array([c.size for c in chunk], dtype=np.int64)
            offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            flat = np.concatenate(chunk) if lens.sum() else np.zeros(1, npdt)
            parts.append((dev(flat, dt), e._i64(offs), e._i64(lens)))
        for q in (0.0, 0.1, 0.5, 1.0):
            ranks = [int(round(q * (len(v) - 1))) for v in seg_vals]
            got = host(ck.order_statistics(e, parts, [ranks], dt, ck.LocalReducer())[0])
            want = np.array([np.sort(v)[r] for v, r in zip(seg_vals, ranks)], dtype=np.float64)
            assert np.array_equal(got, want), (q, got, want)
        centers = dev(np.array([0.3, -0.2, 0.1, 0.0]), torch.float64)
        ranks = [len(v) // 2 for v in seg_vals]
        got = host(ck.order_statistics(e, parts, [ranks], dt, ck.LocalReducer(), absdev=True, centers=centers)[0])
        want = np.array([np.sort(np.abs(v - npdt(c)))[r] for v, r, c in zip(seg_vals, ranks, host(centers))], dtype=np.float64)
        assert np.array_equal(got, want), (got, want)
