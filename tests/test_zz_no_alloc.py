"""The entry point that allocates nothing (adn_wavelet_denoise_noalloc_f32) against the ordinary
one (adn_wavelet_denoise_f32), through the C ABI: same kernels, workspace in the library's static
scratch array.  Two calls in a row, the second on a smaller input (the array keeps the first
call's contents), results bit for bit equal; inputs beyond the scratch are rejected."""
import ctypes as C

import numpy as np
import pytest
import torch

import synth
from gpu_util import engine, dev

pytestmark = pytest.mark.gpu
SR = 48000


def _call(e, fn, x, off, frames, out, n_ch, wavelet, level, max_frames, axis=0):
    nb = len(frames)
    d_off, d_fr = e._i64(off), e._i64(frames)
    sigma = torch.full((nb,), -1.0, dtype=torch.float64, device="cuda:0")
    thresh = torch.full((nb,), -1.0, dtype=torch.float64, device="cuda:0")
    zeroed = torch.full((nb,), -1, dtype=torch.int64, device="cuda:0")
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    e.use_current_stream()
    rc = fn(e._h, p(x), p(out), nb, p(d_off), p(d_fr), p(d_off), int(max_frames), n_ch, axis, wavelet, level, 0, 0,
            p(sigma), p(thresh), p(zeroed))
    return rc, sigma, thresh, zeroed


def _compare(e, x, off, frames, n_ch, wavelet, level):
    lib = e.lib
    mx = int(max(frames))
    size = int(x.numel()) + 2 * len(frames) * n_ch
    ref = torch.zeros(size, dtype=torch.float32, device="cuda:0")
    got = torch.zeros(size, dtype=torch.float32, device="cuda:0")
    rc, s0, t0, z0 = _call(e, lib.adn_wavelet_denoise_f32, x, off, frames, ref, n_ch, wavelet, level, mx)
    assert rc == 0, lib.adn_last_error()
    before = torch.cuda.memory_allocated(), e.workspace_bytes()
    rc, s1, t1, z1 = _call(e, lib.adn_wavelet_denoise_noalloc_f32, x, off, frames, got, n_ch, wavelet, level, mx)
    assert rc == 0, lib.adn_last_error()
    torch.cuda.synchronize()
    assert e.workspace_bytes() == before[1]                    # the growable workspace was not touched
    assert torch.equal(ref.view(torch.int32), got.view(torch.int32))      # bit for bit, NaNs included
    assert torch.equal(s0, s1) and torch.equal(t0, t1) and torch.equal(z0, z1)
    assert float(s1.min()) > 0.0 and int(z1.min()) > 0


def test_no_alloc_entry_point_matches_ordinary_one():
    e = engine()
    # first call: one 30 s mono clip in 10 blocks + an 11th short odd block, db8 level 8 (tile kernels,
    # deep levels, the long-segment statistics)
    n = 30 * SR + 4801
    x = dev(synth.noisy_speech(n, SR, 42, 900).astype(np.float32))
    bs = 144000
    starts = np.arange(0, n, bs, dtype=np.int64)
    lens = np.minimum(bs, n - starts)
    even_off = np.concatenate([[0], np.cumsum(lens + (lens & 1))[:-1]])
    # the ordinary entry point reads blocks at `starts`; reuse starts as output offsets of an even-padded copy
    xp = torch.zeros(int((lens + (lens & 1)).sum()), dtype=torch.float32, device="cuda:0")
    for s, l, o in zip(starts, lens, even_off):
        xp[o:o + l] = x[s:s + l]
    _compare(e, xp, even_off, lens, 1, 1, 8)
    # second call, smaller input, different shape of work: 3 s of stereo (single-pass statistics) ...
    xs = dev(synth.stereo_noisy_speech(3 * SR, SR, 42, 901).astype(np.float32)).view(-1)
    fr = np.full(10, 3 * SR // 10, dtype=np.int64)
    _compare(e, xs, np.arange(10, dtype=np.int64) * fr[0] * 2, fr, 2, 0, 2)
    # ... and a third, tiny mono db4 call (short-segment statistics)
    xt = dev(synth.noisy_speech(8000, 16000, 42, 902).astype(np.float32))
    _compare(e, xt, np.arange(4, dtype=np.int64) * 2000, np.full(4, 2000, dtype=np.int64), 1, 0, 2)


def test_no_alloc_entry_point_rejects_what_does_not_fit():
    from adn_hip import _capi
    e = engine()
    lib = e.lib
    n = 40 * 1000 * 1000                         # 160 MB of samples: the statistics alone exceed the scratch
    x = torch.zeros(n, dtype=torch.float32, device="cuda:0")
    out = torch.empty_like(x)
    rc, *_ = _call(e, lib.adn_wavelet_denoise_noalloc_f32, x, [0], [n], out, 1, 0, 2, n)
    assert rc == _capi.ERR_NOMEM
    msg = lib.adn_last_error().decode()
    assert "allocates nothing" in msg and str(_capi.NOALLOC_SCRATCH_BYTES) in msg
    # the time-axis mode of multi-channel blocks needs a second buffer: rejected as well
    xs = torch.zeros(2 * 4800, dtype=torch.float32, device="cuda:0")
    rc, *_ = _call(e, lib.adn_wavelet_denoise_noalloc_f32, xs, [0], [4800], torch.empty_like(xs), 2, 0, 2, 4800, axis=1)
    assert rc == _capi.ERR_NOMEM
    # and the ordinary entry point is unaffected by the failed calls
    y = e.denoise_signal(dev(synth.noisy_speech(16000, 16000, 42, 903).astype(np.float32)), 16000)
    assert torch.isfinite(y).all()
