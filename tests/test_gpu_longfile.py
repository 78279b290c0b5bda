"""GPU parity on files of the length the reference was written for.  The only run the reference
records (logs/phase2_wavelet.json:6-7, logs/core_phase2c_wiener_stdout.log:3) is a 484.6 s stereo
file at 44.1 kHz: every STFT row then has more than 32 K frames and every 10 % block of the wavelet
stage more than a million frames.  No diagnostic option is set here: the kernels are the ones the
sizes select by themselves."""
import numpy as np
import pytest
import torch

import adn_oracle as orc
import synth
from gpu_util import engine, dev, host, check64, check32

pytestmark = pytest.mark.gpu
SR = 44100


def _as64(t):
    return host(t).astype(np.float64)


@pytest.mark.timeout(400)
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_long_stereo_file_chain(dtype):
    """One 400 s stereo file at 44.1 kHz (34 452 STFT frames per row, blocks of 1 764 000 frames) and a
    5 s noise sample through noise_psd -> denoise_signal -> spectral_subtract -> wiener (voice-activity
    guidance on and off); every stage against the oracle on the input the stage saw on the GPU."""
    e = engine()
    dt = torch.float32 if dtype == "f32" else torch.float64
    chk = check32 if dtype == "f32" else check64
    n, nn = 400 * SR, 5 * SR
    xd = torch.empty((n, 2), dtype=dt, device="cuda:0")
    xd[:, 0] = e.synth_noisy_speech(n, SR, stream=3000).to(dt)
    xd[:, 1] = e.synth_noisy_speech(n, SR, stream=3001, speech_gain=0.8).to(dt)
    nd = dev(synth.white_noise(nn, 42, 103000), dt)
    assert e.stft_num_frames(n, 2048, 512) > 32768 and e.denoise_block_frames(SR, n) > 1000000

    base = e.noise_psd(nd, [nn])
    b_or, _ = orc.noise_profile(_as64(nd))
    assert np.max(np.abs(_as64(base[0]) - b_or) / b_or) <= (3e-6 if dtype == "f32" else 1e-12)

    y2, (sigma, thresh, zeroed) = e.denoise_signal(xd, SR, want_stats=True)
    y2_or, s_or, t_or = orc.denoise_signal(_as64(xd), SR)
    chk(_as64(y2), y2_or)
    assert np.max(np.abs(host(sigma) - s_or) / s_or) <= (1e-6 if dtype == "f32" else 1e-12)
    del y2_or

    b32 = _as64(base[0]).astype(np.float32).astype(np.float64)
    y2b, l2b = e.spectral_subtract(y2.view(-1), [n], dev(b32, dt), n_ch=2)
    y2b_or = orc.spectral_subtract(_as64(y2), b32)
    assert int(l2b[0]) == y2b_or.size
    chk(_as64(y2b), y2b_or)
    del y2b_or

    x2c = _as64(y2b)
    for vad in (True, False):
        y2c, l2c, gain = e.wiener(y2b, [int(l2b[0])], SR, vad_guided=vad, want_gain=True)
        y2c_or, g_or = orc.wiener(x2c, SR, vad_guided=vad)
        assert int(l2c[0]) == y2c_or.size
        chk(_as64(y2c), y2c_or)
        assert np.max(np.abs(_as64(gain[0]) - g_or)) <= (2e-4 if dtype == "f32" else 1e-10)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("frames", [8200, 11500, 23000])
def test_mono_rows_above_8192_frames(dtype, frames):
    """Mono files whose rows are just above 8192 frames (the longest row one wave holds in registers)
    and above 11 264 and 22 528 (where the row select takes more waves per row): Wiener stage with and
    without voice-activity guidance, the file as a noise sample through the noise profile, and the
    spectral subtraction, against the oracle."""
    e = engine()
    dt = torch.float32 if dtype == "f32" else torch.float64
    chk = check32 if dtype == "f32" else check64
    n = (frames - 1) * 512 + 2048 - 100                   # a last, zero-padded frame
    xd = e.synth_noisy_speech(n, SR, stream=3100 + frames).to(dt)
    assert e.stft_num_frames(n, 2048, 512) == frames
    x64 = _as64(xd)
    for vad in (True, False):
        y, _, gain = e.wiener(xd, [n], SR, vad_guided=vad, want_gain=True)
        want, g_or = orc.wiener(x64, SR, vad_guided=vad)
        chk(_as64(y), want)
        assert np.max(np.abs(_as64(gain[0]) - g_or)) <= (2e-4 if dtype == "f32" else 1e-10)
    base = _as64(e.noise_psd(xd, [n])[0])
    b_or = orc.noise_profile(x64)[0]
    assert np.max(np.abs(base - b_or)) <= (2e-6 if dtype == "f32" else 1e-12) * max(1.0, float(np.max(np.abs(b_or))))
    nm = np.abs(np.sin(np.arange(1025) * 0.013)) * 0.004 + 0.002
    nm = nm.astype(np.float32).astype(np.float64)
    yb, _ = e.spectral_subtract(xd, [n], dev(nm, dt))
    chk(_as64(yb), orc.spectral_subtract(x64, nm))
