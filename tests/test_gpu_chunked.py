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


@pytest.mark.parametrize("dtype,n_fft,hop", [("f32", 2048, 512), ("f64", 1024, 256)])
def test_spectral_subtract_in_chunks_smooth_time_5(dtype, n_fft, hop):
    """A time smoothing of five frames (the reference reads the count from its configuration,
    src/spectral_subtract.py:58-59) reads two frames on either side: the chunks' halo must grow with it
    (reach = 2), else the hops next to a seam differ from the one-call result.  (binary64 at n_fft 2048 has no
    room in LDS for five frames of smoothing: that precision runs at n_fft 1024 here.)"""
    e = engine()
    dt = torch.float64 if dtype == "f64" else torch.float32
    n = 12 * SR + 333
    x = dev(synth.noisy_speech(n, SR, 42, 912), dt)
    prof = e.noise_psd(dev(synth.white_noise(2 * SR, 42, 913), dt), [2 * SR], n_fft=n_fft, hop=hop)[0]
    kw = dict(smooth_time=5, n_fft=n_fft, hop=hop)
    y1, l1 = e.spectral_subtract(x, [n], prof, **kw)
    hops, frames = ck.plan_stft_chunks(n, 3, n_fft, hop)
    pieces, short = [], []
    for h0, h1 in hops:
        s0, s1, keep = ck.stft_chunk_window(h0, h1, frames, n, n_fft, hop, reach=2)
        pieces.append(ck.spectral_subtract_chunk(e, x[s0:s1].contiguous(), keep, (h1 - h0) * hop, prof, **kw))
        s0, s1, keep = ck.stft_chunk_window(h0, h1, frames, n, n_fft, hop, reach=1)
        short.append(ck.spectral_subtract_chunk(e, x[s0:s1].contiguous(), keep, (h1 - h0) * hop, prof, **kw))
    assert torch.equal(torch.cat(pieces), y1)
    assert not torch.equal(torch.cat(short), y1)            # the halo of the 3-frame smoothing is too short
    want = orc.spectral_subtract(host(x).astype(np.float64), host(prof).astype(np.float64), smooth_time=5, n_fft=n_fft, hop=hop)
    (check64 if dtype == "f64" else check32)(host(y1), want)
