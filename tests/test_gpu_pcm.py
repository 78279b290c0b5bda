"""PCM sample conversion on the device: the standalone kernel and the fused store of the STFT
stages against the sample-at-a-time restatement (bit-exact: integer work)."""
import numpy as np
import pytest
import torch

import adn_oracle as orc
from gpu_util import engine, dev, host

pytestmark = pytest.mark.gpu


def _edge_values():
    rng = np.random.default_rng(9)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.5, -1.5, 0.99999994, -0.99999994, 1 - 2.0 ** -16, 2.0 ** -16,
                     -2.0 ** -16, 3 * 2.0 ** -17, -3 * 2.0 ** -17, 0.5 * 2.0 ** -31, 1.5 * 2.0 ** -31,
                     2.5 * 2.0 ** -31, -0.5 * 2.0 ** -31, np.nan, np.inf, -np.inf, 1e30, -1e30,
                     (65536 * 7 - 0.5) / 2.0 ** 31, -(65536 * 7 + 0.5) / 2.0 ** 31])
    return np.concatenate([edge, rng.uniform(-1.1, 1.1, 5003), rng.standard_normal(1000) * 1e-4])


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_pcm_encode_bit_exact(dt):
    e = engine()
    x = _edge_values().astype(np.float32 if dt == torch.float32 else np.float64)
    for n in (x.size, 1, 3, 1024, 0):
        for bits in (16, 24, 32):
            xs = x[:n]
            q, xq = e.pcm_encode(dev(xs, dt), bits, dequantize=True)
            want = orc.pcm_encode(xs.astype(np.float64), bits)
            assert q.dtype == (torch.int16 if bits == 16 else torch.int32)
            assert np.array_equal(host(q).astype(np.int64), want), (n, bits)
            assert np.array_equal(host(xq).astype(np.float64), orc.pcm_decode(want, bits).astype(xs.dtype))
    xd = dev(x, dt)
    q, same = e.pcm_encode(xd, 16, dequantize="inplace")
    assert same.data_ptr() == xd.data_ptr()
    assert np.array_equal(host(xd).astype(np.float64), orc.pcm_decode(orc.pcm_encode(x.astype(np.float64)), 16))
    with pytest.raises(Exception, match="bits must be"):
        e.pcm_encode(xd, 8)


@pytest.mark.parametrize("dt,n_fft,hop", [(torch.float32, 2048, 512), (torch.float64, 2048, 512),
                                           (torch.float32, 512, 128)])
def test_stage_store_writes_file_samples(dt, n_fft, hop):
    """The optional PCM output of phases 2b / 2c equals encoding the plain output afterwards, and
    the float output is what a reader of the 16-bit file gets."""
    e = engine()
    sr = 16000
    lens = [40000, 23456, 5000]
    x = torch.cat([e.synth_noisy_speech(n, sr, stream=i, dtype=dt, speech_gain=3.0) for i, n in enumerate(lens)]) * 4.0
    noise = dev(np.full(n_fft // 2 + 1, 0.05), dt)
    for name in ("spectral_subtract", "wiener"):
        if name == "spectral_subtract":
            run = lambda **kw: e.spectral_subtract(x, lens, noise, n_fft=n_fft, hop=hop, **kw)[:2]
        else:
            run = lambda **kw: e.wiener(x, lens, sr, n_fft=n_fft, hop=hop, **kw)[:2]
        y, out_lens = run()
        pcm = torch.full((y.numel(),), 12345, dtype=torch.int16, device=y.device)
        yq, _ = run(pcm16=pcm)
        want = orc.pcm_encode(host(y).astype(np.float64), 16)
        assert np.array_equal(host(pcm).astype(np.int64), want), name
        assert np.array_equal(host(yq).astype(np.float64), orc.pcm_decode(want, 16)), name
        assert np.abs(host(y)).max() == 1.0           # the clip is exercised
    with pytest.raises(ValueError, match="pcm16 must be"):
        e.wiener(x, lens, sr, pcm16=torch.zeros(10, dtype=torch.int16, device=x.device))
