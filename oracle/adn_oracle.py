"""CPU restatement (numpy only) of the reference's core denoising DSP chain.

THIS FILE IS TEST INFRASTRUCTURE.  It is the checker for the HIP path: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product path never routes through it.

Parity status: pinned.  The reference holds no golden vectors of its own for
this path, so every function below is pinned against outputs of the reference
itself (run in the build container with PyWavelets 1.1.1 / SciPy 1.7.1 /
NumPy 1.26.4) that are committed under ``tests/golden/`` together with the
script that produced them (``oracle/gen_golden.py``).

The wavelet arithmetic lives in a third-party dependency of the reference that
is not vendored there (PyWavelets, unpinned by the reference); its published
periodized / symmetric filter-bank algorithm is restated here.  The STFT
arithmetic lives in SciPy (``scipy.signal.stft/istft/medfilt``,
``scipy.ndimage.uniform_filter``) and is restated the same way.

Each function cites the reference lines (relative to the reference root) whose
behaviour it restates.
"""
import math

import numpy as np

# --------------------------------------------------------------------------
# Daubechies filter banks (decomposition low-pass taps; the other three filters
# follow from the quadrature-mirror relations).
# --------------------------------------------------------------------------
_DEC_LO = {
    "db4": [-0.010597401785069032, 0.0328830116668852, 0.030841381835560764,
            -0.18703481171909309, -0.027983769416859854, 0.6308807679298589,
            0.7148465705529157, 0.2303778133088965],
    "db8": [-0.00011747678412476953, 0.0006754494064505693, -0.00039174037337694705,
            -0.004870352993451574, 0.008746094047405777, 0.013981027917398282,
            -0.044088253930794755, -0.017369301001807547, 0.12874742662047847,
            0.0004724845739132828, -0.2840155429615469, -0.015829105256349306,
            0.5853546836542067, 0.6756307362972898, 0.31287159091429995,
            0.05441584224310401],
}


def filters(wavelet):
    """(dec_lo, dec_hi, rec_lo, rec_hi) as float64 arrays."""
    h = np.array(_DEC_LO[wavelet], dtype=np.float64)
    F = h.size
    sign = np.where(np.arange(F) % 2 == 0, -1.0, 1.0)      # (-1)^(j+1)
    g = sign * h[::-1]
    return h, g, h[::-1].copy(), g[::-1].copy()


# --------------------------------------------------------------------------
# Single-level transforms along the last axis
# --------------------------------------------------------------------------
def _even_extend(x):
    if x.shape[-1] % 2:
        x = np.concatenate([x, x[..., -1:]], axis=-1)
    return x


def dwt_per(x, wavelet):
    """One periodized analysis level (pywt.dwt(x, w, 'periodization')).

    Called by the reference through pywt.wavedec at denoise.py:77.
    cA[k] = sum_j h[j] * xe[(2k + F/2 - j) mod M], M = even-extended length.
    """
    h, g, _, _ = filters(wavelet)
    F = h.size
    xe = _even_extend(np.asarray(x, dtype=np.float64))
    M = xe.shape[-1]
    K = M // 2
    k = np.arange(K)
    cA = np.zeros(xe.shape[:-1] + (K,))
    cD = np.zeros_like(cA)
    for j in range(F):
        v = xe[..., (2 * k + F // 2 - j) % M]
        cA += h[j] * v
        cD += g[j] * v
    return cA, cD


def idwt_per(cA, cD, wavelet):
    """One periodized synthesis level (pywt.idwt(cA, cD, w, 'periodization')).

    y[(j + 2k - (F/2 - 1)) mod 2K] += rec_lo[j]*cA[k] + rec_hi[j]*cD[k].
    """
    _, _, rl, rh = filters(wavelet)
    F = rl.size
    cA = np.asarray(cA, dtype=np.float64)
    cD = np.asarray(cD, dtype=np.float64)
    K = cA.shape[-1]
    y = np.zeros(cA.shape[:-1] + (2 * K,))
    k = np.arange(K)
    for j in range(F):
        idx = (j + 2 * k - (F // 2 - 1)) % (2 * K)
        # for K >= F/2 idx is a permutation of a subset; for tiny K it can repeat
        np.add.at(y, (Ellipsis, idx), rl[j] * cA + rh[j] * cD)
    return y


def dwt_sym(x, wavelet):
    """One analysis level with half-sample symmetric extension (pywt 'symmetric').

    Reached in the reference through pywt.WaveletPacket(data, 'db8', 'symmetric', 4)
    at lib/windowBundle.py:33-34.  cA[k] = sum_j h[j]*xs[2k + 1 - j],
    k = 0 .. floor((N+F-1)/2)-1, xs[-1-i] = x[i], xs[N+i] = x[N-1-i].
    """
    h, g, _, _ = filters(wavelet)
    F = h.size
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[-1]
    K = (N + F - 1) // 2
    k = np.arange(K)
    cA = np.zeros(x.shape[:-1] + (K,))
    cD = np.zeros_like(cA)
    for j in range(F):
        i = 2 * k + 1 - j
        # reflect (half-sample symmetric), repeatedly for signals shorter than the filter
        i = np.mod(i, 2 * N)
        i = np.where(i >= N, 2 * N - 1 - i, i)
        v = x[..., i]
        cA += h[j] * v
        cD += g[j] * v
    return cA, cD


def idwt_sym(cA, cD, wavelet):
    """Inverse of dwt_sym (pywt.idwt(..., 'symmetric')): output length 2K - F + 2."""
    _, _, rl, rh = filters(wavelet)
    F = rl.size
    cA = np.asarray(cA, dtype=np.float64)
    cD = np.asarray(cD, dtype=np.float64)
    K = cA.shape[-1]
    n_out = 2 * K - F + 2
    full = np.zeros(cA.shape[:-1] + (2 * K + F - 1,))
    for j in range(F):
        full[..., j:j + 2 * K:2] += rl[j] * cA + rh[j] * cD
    return full[..., F - 2:F - 2 + n_out]


# --------------------------------------------------------------------------
# Multi-level transforms
# --------------------------------------------------------------------------
def wavedec(x, wavelet="db4", level=2, mode="per"):
    """[cA_L, cD_L, ..., cD_1] like pywt.wavedec (denoise.py:77)."""
    step = dwt_per if mode == "per" else dwt_sym
    a = np.asarray(x, dtype=np.float64)
    details = []
    for _ in range(level):
        a, d = step(a, wavelet)
        details.append(d)
    return [a] + details[::-1]


def waverec(coeffs, wavelet="db4", mode="per"):
    """pywt.waverec (denoise.py:89) including its trim-by-one rule."""
    step = idwt_per if mode == "per" else idwt_sym
    a = coeffs[0]
    for d in coeffs[1:]:
        if a.shape[-1] == d.shape[-1] + 1:
            a = a[..., :-1]
        elif a.shape[-1] != d.shape[-1]:
            raise ValueError("coefficient shape mismatch")
        a = step(a, d, wavelet)
    return a


def band_lengths(n, level, wavelet="db4", mode="per"):
    """Lengths [cA_L, cD_L, ..., cD_1] for an input of n samples."""
    F = len(_DEC_LO[wavelet])
    out = []
    for _ in range(level):
        n = (n + 1) // 2 if mode == "per" else (n + F - 1) // 2
        out.append(n)
    return [out[-1]] + out[::-1]


# --------------------------------------------------------------------------
# VisuShrink block denoise
# --------------------------------------------------------------------------
def mad(arr):
    """median(|a - median(a)|) over the flattened array, no 0.6745 (denoise.py:16-23)."""
    a = np.asarray(arr, dtype=np.float64).ravel()
    med = np.median(a)
    return np.median(np.abs(a - med))


def threshold(c, t, mode="soft"):
    """pywt.threshold (denoise.py:86): soft = c*max(1 - t/|c|, 0) with IEEE 0/0 -> NaN."""
    c = np.asarray(c, dtype=np.float64)
    if mode == "soft":
        with np.errstate(divide="ignore", invalid="ignore"):
            mag = np.abs(c)
            f = 1.0 - t / mag
            f = np.where(f < 0.0, 0.0, f)       # clip(min=0) keeps NaN
            return c * f
    if mode == "hard":
        return np.where(np.abs(c) < t, 0.0, c)
    raise ValueError(mode)


def denoise_block(block, wavelet="db4", level=2, thr_mode="soft", return_bands=False, mode="per"):
    """One iteration of the reference block loop (denoise.py:77-89).

    The transform runs along the LAST axis, so a (frames, channels) block is
    transformed across channels, exactly as the reference does.  `mode` is the
    reference's 'per'; 'sym' is pywt's 'symmetric' extension (1-D blocks only).
    Returns (clean, sigma, thresh).
    """
    block = np.asarray(block, dtype=np.float64)
    coeffs = wavedec(block, wavelet, level, mode)
    sigma = mad(coeffs[-1])
    with np.errstate(divide="ignore"):
        thresh = sigma * np.sqrt(2 * np.log(len(block)))
    shr = [coeffs[0]] + [threshold(c, thresh, thr_mode) for c in coeffs[1:]]
    clean = waverec(shr, wavelet, mode)
    if return_bands:
        return clean, sigma, thresh, coeffs
    return clean, sigma, thresh


def denoise_block_size(rate, frames):
    """Frames per block: int(rate * duration * 0.10), duration = frames/rate (denoise.py:76)."""
    duration = float(frames) / rate
    return int(rate * duration * 0.10)


def denoise_signal(x, rate, wavelet="db4", level=2, thr_mode="soft"):
    """AudioDeNoise.deNoise on an in-memory signal (denoise.py:59-90).

    x: (frames,) mono or (frames, channels).  Returns (y, sigmas, threshs).
    y can be longer than x when a block has odd length (periodized padding).
    """
    x = np.asarray(x, dtype=np.float64)
    bs = denoise_block_size(rate, x.shape[0])
    if bs <= 0:
        raise ValueError("input shorter than 10 frames: block size is zero")
    outs, sig, thr = [], [], []
    for s in range(0, x.shape[0], bs):
        c, sg, th = denoise_block(x[s:s + bs], wavelet, level, thr_mode)
        outs.append(c)
        sig.append(sg)
        thr.append(th)
    return np.concatenate(outs, axis=0), np.array(sig), np.array(thr)


# --------------------------------------------------------------------------
# STFT / ISTFT exactly as the reference calls SciPy
# --------------------------------------------------------------------------
def hann_periodic(n):
    """scipy.signal.get_window('hann', n): 0.5 - 0.5 cos(2 pi k / n)."""
    k = np.arange(n)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def stft_frames(n, n_fft, hop):
    """(n_pad, T) for scipy stft(boundary=None, padded=True)."""
    if n < n_fft:
        raise ValueError("window is longer than input signal")
    nadd = (-(n - n_fft) % hop) % n_fft
    return n + nadd, 1 + (n + nadd - n_fft) // hop


def stft(x, n_fft, hop, window=None, padded=True):
    """scipy.signal.stft(x, nperseg=n_fft, noverlap=n_fft-hop, boundary=None, padded=...)

    src/spectral_subtract.py:82, src/wiener_filter.py:72, src/noise_profile.py:13.
    Returns Z with shape (n_fft/2+1, T).
    """
    x = np.asarray(x, dtype=np.float64)
    w = hann_periodic(n_fft) if window is None else np.asarray(window, dtype=np.float64)
    if x.shape[-1] < n_fft:
        raise ValueError("window is longer than input signal")
    if padded:
        n_pad, T = stft_frames(x.shape[-1], n_fft, hop)
        x = np.concatenate([x, np.zeros(n_pad - x.shape[-1])])
    else:
        T = 1 + (x.shape[-1] - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(T)[:, None]
    seg = x[idx] * w[None, :]
    Z = np.fft.rfft(seg, n=n_fft, axis=1) * np.sqrt(1.0 / w.sum() ** 2)
    return Z.T


def stft_axes(n, sr, n_fft, hop, padded=True):
    """(frequencies, times) that scipy.signal.stft returns next to Z."""
    n_pad = stft_frames(n, n_fft, hop)[0] if padded else n
    t = np.arange(n_fft / 2, n_pad - n_fft / 2 + 1, hop) / float(sr)
    f = np.fft.rfftfreq(n_fft, 1.0 / sr)
    return f, t


def istft(Z, n_fft, hop, window=None):
    """scipy.signal.istft(Z, nperseg=n_fft, noverlap=n_fft-hop) with default boundary=True.

    src/spectral_subtract.py:108, src/wiener_filter.py:128.  Output length (T-1)*hop.
    """
    w = hann_periodic(n_fft) if window is None else np.asarray(window, dtype=np.float64)
    T = Z.shape[1]
    seg = np.fft.irfft(Z, n=n_fft, axis=0) * w.sum()          # (n_fft, T)
    out = np.zeros(n_fft + (T - 1) * hop)
    norm = np.zeros_like(out)
    for t in range(T):
        out[t * hop:t * hop + n_fft] += seg[:, t] * w
        norm[t * hop:t * hop + n_fft] += w ** 2
    half = n_fft // 2
    out = out[half:-half]
    norm = norm[half:-half]
    return out / np.where(norm > 1e-10, norm, 1.0)


def box_filter_nearest(M, size, axis):
    """scipy.ndimage.uniform_filter1d(M, size, axis, mode='nearest') (edge-replicated mean)."""
    M = np.asarray(M, dtype=np.float64)
    left = size // 2
    right = size - left - 1
    n = M.shape[axis]
    acc = np.zeros_like(M)
    for o in range(-left, right + 1):
        idx = np.clip(np.arange(n) + o, 0, n - 1)
        acc += np.take(M, idx, axis=axis)
    return acc / size


def smooth_2d(M, tf=3, ff=5):
    """uniform_filter(M, size=(ff, tf), mode='nearest') (src/noise_profile.py:16-19,
    src/spectral_subtract.py:25-27): axis 0 = frequency, axis 1 = time."""
    return box_filter_nearest(box_filter_nearest(M, ff, 0), tf, 1)


def medfilt(v, k):
    """scipy.signal.medfilt(v, k) for 1-D v: running median with zero padding."""
    v = np.asarray(v, dtype=np.float64)
    h = k // 2
    p = np.concatenate([np.zeros(h), v, np.zeros(h)])
    win = np.stack([p[i:i + v.size] for i in range(k)], axis=0)
    return np.sort(win, axis=0)[h]


# --------------------------------------------------------------------------
# Phase 1: noise profile (src/noise_profile.py:43-75)
# --------------------------------------------------------------------------
def fold_mono(x):
    x = np.asarray(x, dtype=np.float64)
    return np.mean(x, axis=1) if x.ndim > 1 else x


def noise_profile(noise, n_fft=2048, hop=512, win_length=2048, percentile=95,
                  spectral_floor=0.1, time_frames=5, freq_bins=5, padded=True):
    """Returns baseline[F] (float64, before the float32 cast) and T.

    The reference builds an F x T matrix that is constant along time
    (outer(global_noise, ones)), clamps and box-smooths it, and stores it as
    float32; only column 0 carries information.
    """
    n = fold_mono(noise)
    N = np.abs(stft(n, win_length, hop, hann_periodic(win_length), padded))
    global_noise = np.percentile(N, percentile, axis=1)
    base = np.maximum(global_noise, spectral_floor * np.max(N))
    base = box_filter_nearest(base, freq_bins, 0)
    # time smoothing of a time-constant matrix: a no-op up to rounding
    return base, N.shape[1]


# --------------------------------------------------------------------------
# Phase 2b: spectral subtraction (src/spectral_subtract.py:29-46, 75-109)
# --------------------------------------------------------------------------
def resize_noise(nf, F):
    """np.interp resize used when the profile's bin count differs (spectral_subtract.py:91-94)."""
    nf = np.asarray(nf, dtype=np.float64)
    return np.interp(np.linspace(0, len(nf) - 1, F), np.arange(len(nf)), nf)


def spectral_subtract_spec(Z, noise_mag, alpha=2.2, beta=0.1, smooth_time=3, smooth_freq=5):
    mag = np.abs(Z)
    noise_mag = np.asarray(noise_mag, dtype=np.float64).reshape(-1, 1)
    clean = np.maximum(mag - alpha * noise_mag, beta * mag)
    clean_s = smooth_2d(clean, tf=smooth_time, ff=smooth_freq)
    return clean_s * np.exp(1j * np.angle(Z))


def spectral_subtract(x, noise_mag, n_fft=2048, hop=512, alpha=2.2, beta=0.1,
                      smooth_time=3, smooth_freq=5):
    """x mono or multi-channel -> y mono of length (T-1)*hop, clipped to [-1, 1]."""
    x = fold_mono(x)
    Z = stft(x, n_fft, hop)
    noise_mag = np.asarray(noise_mag, dtype=np.float64)
    if noise_mag.shape[0] != Z.shape[0]:
        noise_mag = resize_noise(noise_mag, Z.shape[0])
    Zc = spectral_subtract_spec(Z, noise_mag, alpha, beta, smooth_time, smooth_freq)
    return np.clip(istft(Zc, n_fft, hop), -1.0, 1.0)


# --------------------------------------------------------------------------
# Phase 2c: Wiener filter (src/wiener_filter.py:12-39, 66-129)
# --------------------------------------------------------------------------
def energy_vad(x, sr, frame_ms=30, threshold_db=-40, hangover_frames=5):
    """src/wiener_filter.py:12-33.  Returns (bool mask per VAD frame, frame_len)."""
    x = np.asarray(x, dtype=np.float64)
    frame_len = int(sr * frame_ms / 1000)
    n_frames = max(1, int(np.ceil(len(x) / frame_len)))
    speech = np.zeros(n_frames, dtype=bool)
    for i in range(n_frames):
        fr = x[i * frame_len:min(len(x), (i + 1) * frame_len)]
        if fr.size == 0:
            continue
        rms = np.sqrt(np.mean(fr ** 2) + 1e-12)
        speech[i] = 20 * np.log10(rms + 1e-12) > threshold_db
    out = speech.copy()
    last_true = -999
    for i in range(n_frames):
        if speech[i]:
            last_true = i
        if i - last_true <= hangover_frames:
            out[i] = True
    return out, frame_len


def vad_per_stft_frame(vad_mask, frame_len, T, hop):
    """src/wiener_filter.py:92-98."""
    i = np.arange(T)
    idx = np.minimum((i * hop / frame_len).astype(np.int64), len(vad_mask) - 1)
    return vad_mask[idx]


def wiener_gain(x, sr, n_fft=2048, hop=512, update_ms=20, kernel_bins=5, vad_guided=True):
    """Per-bin gain of the reference Wiener stage plus the STFT it was derived from."""
    Z = stft(x, n_fft, hop)
    S = np.abs(Z) ** 2
    noise_psd = np.maximum(medfilt(np.percentile(S, 10, axis=1), kernel_bins), 1e-12)
    if vad_guided:
        mask, frame_len = energy_vad(x, sr, frame_ms=update_ms, threshold_db=-40, hangover_frames=5)
        m = vad_per_stft_frame(mask, frame_len, Z.shape[1], hop)
        speech = np.sum(np.where(m[None, :], S, 0.0), axis=1) / np.maximum(np.sum(m), 1)
    else:
        speech = np.median(S, axis=1)
    speech = np.maximum(medfilt(speech, kernel_bins), noise_psd)
    snr = speech / (noise_psd + 1e-12)
    gain = medfilt(np.clip(snr / (1.0 + snr), 0.0, 1.0), kernel_bins)
    return gain, Z


def wiener(x, sr, n_fft=2048, hop=512, update_ms=20, kernel_bins=5, vad_guided=True):
    """x mono or multi-channel -> (y, gain); the dead attack/release loop of
    src/wiener_filter.py:118-125 never runs, so the gain is constant in time."""
    x = fold_mono(x)
    gain, Z = wiener_gain(x, sr, n_fft, hop, update_ms, kernel_bins, vad_guided)
    y = np.clip(istft(Z * gain[:, None], n_fft, hop), -1.0, 1.0)
    return y, gain


# --------------------------------------------------------------------------
# RMS-window noise profiler (lib/noiseProfiler.py, lib/windowBundle.py, lib/linkedList.py)
# --------------------------------------------------------------------------
def band_summary(f, Z, bands):
    """src/analyze_audio.py:50-62 summarize_band_energy: per band (inclusive frequency range) the dB of
    the mean magnitude and the 95th / 50th percentile of the dB magnitudes."""
    mag = np.abs(Z)
    out = {}
    for name, (lb, ub) in bands.items():
        band = mag[(f >= lb) & (f <= ub), :]
        db = 20 * np.log10(band + 1e-12)
        out[name] = {"mean_db": float(20 * np.log10(np.mean(band) + 1e-12)),
                     "p95_db": float(np.percentile(db, 95)), "p50_db": float(np.percentile(db, 50))}
    return out


def analysis_report(x, sr_x, noise, sr_n, bands, n_fft=2048, hop=512, padded=True, seed=42,
                    vad_enabled=True, frame_ms=30):
    """The JSON report of src/analyze_audio.py:113-135, 160-170 (plots excluded)."""
    x, noise = fold_mono(np.asarray(x, dtype=np.float64)), fold_mono(np.asarray(noise, dtype=np.float64))
    Zx, Zn = stft(x, n_fft, hop, padded=padded), stft(noise, n_fft, hop, padded=padded)
    fx = stft_axes(len(x), sr_x, n_fft, hop, padded)[0]
    fn = stft_axes(len(noise), sr_n, n_fft, hop, padded)[0]
    frame_len = max(int(sr_x * frame_ms / 1000), 1)
    return {"seed": seed, "samplerate_interview": sr_x, "samplerate_noise": sr_n,
            "child_band_summary_interview_db": band_summary(fx, Zx, bands),
            "child_band_summary_noise_db": band_summary(fn, Zn, bands),
            "noise_profile_time_mean_mag": np.mean(np.abs(Zn), axis=0).tolist(),
            "vad_frames": int(len(x) / frame_len) if vad_enabled else 0}


def median_filter_reflect(S, size, axis):
    """scipy.ndimage.median_filter(S, size=(1, size)) / ((size, 1)), mode='reflect' (edge sample
    repeated: d c b a | a b c d | d c b a), odd size.  src/hpss_separate.py:54, 57."""
    S = np.asarray(S, dtype=np.float64)
    h = size // 2
    M = np.moveaxis(S, axis, -1)
    n = M.shape[-1]
    idx = np.arange(-h, n + h)
    period = 2 * n
    idx = np.mod(idx, period)
    idx = np.where(idx >= n, period - 1 - idx, idx)
    ext = M[..., idx]
    win = np.lib.stride_tricks.sliding_window_view(ext, size, axis=-1)
    out = np.median(win, axis=-1)
    return np.moveaxis(out, -1, axis)


def hpss_masks(S, margin_h=1.0, margin_p=5.0, kernel_size=31, power=2.0):
    """src/hpss_separate.py:41-67 median_filtering_hpss."""
    Sh = median_filter_reflect(S, kernel_size, 1)
    Sp = median_filter_reflect(S, kernel_size, 0)
    mh = (Sh * margin_h) ** power
    mp = (Sp * margin_p) ** power
    total = mh + mp + 1e-10
    return mh / total, mp / total


def hpss_multiband_masks(S, f, bands, margin_h=1.0, margin_p=5.0, kernel_size=31):
    """src/hpss_separate.py:86-113 multiband_hpss (power stays at its default 2.0 there)."""
    mask_h, mask_p = np.zeros_like(S), np.zeros_like(S)
    for lo, hi in bands:
        bm = (f >= lo) & (f <= hi)
        if not np.any(bm):
            continue
        a, b = hpss_masks(S[bm, :], margin_h, margin_p, kernel_size)
        mask_h[bm, :] = a
        mask_p[bm, :] = b
    return mask_h, mask_p


def hpss_apply_mask_type(mask_h, mask_p, mask_type="soft"):
    """src/hpss_separate.py:126-135."""
    if mask_type == "binary":
        return (mask_h > 0.5).astype(float), (mask_p > 0.5).astype(float)
    if mask_type == "hard":
        mh = (mask_h > mask_p).astype(float)
        return mh, 1.0 - mh
    return mask_h, mask_p


def hpss_local_energy(sig, window_size=2048, hop=512):
    """src/hpss_separate.py:156-161."""
    return np.array([np.sqrt(np.mean(sig[i:i + window_size] ** 2)) for i in range(0, len(sig) - window_size, hop)])


def hpss_preserve_transients(y_h, y_p, y_orig, threshold=0.3, blend=0.7, window_size=2048, hop=512):
    """src/hpss_separate.py:142-190.  The original is NOT shifted by the n_fft/2 the ISTFT drops."""
    m = min(len(y_h), len(y_p), len(y_orig))
    y_h, y_orig = y_h[:m], y_orig[:m]
    eo, eh = hpss_local_energy(y_orig, window_size, hop), hpss_local_energy(y_h, window_size, hop)
    k = min(len(eo), len(eh))
    eo, eh = eo[:k], eh[:k]
    out = y_h.copy()
    if k == 0:
        return out
    tr = (eo - eh) > threshold * np.max(eo)
    for i in np.nonzero(tr)[0]:
        s, e = i * hop, min(i * hop + window_size, m)
        out[s:e] = blend * y_orig[s:e] + (1 - blend) * y_h[s:e]
    return out


def hpss(x, sr, n_fft=2048, hop=512, margin_h=1.0, margin_p=5.0, kernel_size=31, power=2.0, mask_type="soft",
         bands=None, preserve_transients=True):
    """src/hpss_separate.py:233-283: (y_harmonic, y_percussive, metrics) as written to the files / log."""
    x = fold_mono(np.asarray(x, dtype=np.float64))
    Z = stft(x, n_fft, hop)
    S = np.abs(Z)
    if bands is not None:
        f = stft_axes(len(x), sr, n_fft, hop)[0]
        mh, mp = hpss_multiband_masks(S, f, bands, margin_h, margin_p, kernel_size)
    else:
        mh, mp = hpss_masks(S, margin_h, margin_p, kernel_size, power)
    mh, mp = hpss_apply_mask_type(mh, mp, mask_type)
    y_h, y_p = istft(Z * mh, n_fft, hop), istft(Z * mp, n_fft, hop)
    if preserve_transients:
        y_h = hpss_preserve_transients(y_h, y_p, x)
    mx = max(np.max(np.abs(y_h)), np.max(np.abs(y_p)))
    if mx > 0.95:
        y_h, y_p = y_h * (0.95 / mx), y_p * (0.95 / mx)
    rms = lambda v: float(np.sqrt(np.mean(v ** 2)))
    met = {"harmonic_rms": rms(y_h), "percussive_rms": rms(y_p), "total_rms": rms(x)}
    met["harmonic_ratio"] = met["harmonic_rms"] / (met["total_rms"] + 1e-10)
    return y_h, y_p, met


def window_rms(x, win):
    """Per-window sqrt(sum(x^2)/rows) (lib/windowBundle.py:67-74); the sum runs over
    every channel of the window's rows, the divisor is the row count."""
    x = np.asarray(x, dtype=np.float64)
    n_win = math.ceil(x.shape[0] / win)
    out = np.empty(n_win)
    for i in range(n_win):
        w = x[i * win:(i + 1) * win]
        out[i] = math.sqrt(np.sum(np.power(w, 2)) / len(w))
    return out


def noise_windows(rms, percentile_level=95):
    """(threshold, is_noise[]) per lib/noiseProfiler.py:151-162,191-206."""
    order = sorted(range(len(rms)), key=lambda i: rms[i], reverse=True)   # stable, like sorted()
    thr = rms[order[math.floor(percentile_level / 100 * len(rms))]]
    is_noise = np.array([r < (thr + 0.05 * thr) for r in rms], dtype=bool)
    return thr, is_noise


def predicted_noise_order(is_noise):
    """Window indices, in output order, of NoiseProfiler.getNoiseDataPredicted
    (lib/noiseProfiler.py:79-149, lib/linkedList.py:14-30).

    Noise windows are kept in place.  A run of n signal windows is replaced by n
    noise windows: walking backwards from the nearest noise window BEFORE the
    noise window that ends the run (restarting at that window whenever the
    contiguous run of noise windows is exhausted); when no noise window precedes
    it, walking forwards from the nearest noise window AFTER it.  A trailing run
    is predicted the same way relative to the last noise window.
    """
    n = len(is_noise)

    def prev_valid(i):
        return next((j for j in range(i - 1, -1, -1) if is_noise[j]), None)

    def next_valid(i):
        return next((j for j in range(i + 1, n) if is_noise[j]), None)

    def circular(start, count, step):
        if start is None:
            raise ValueError("no noise window available to fill a gap")
        out, cur = [], start
        while True:
            out.append(cur)
            if len(out) == count:
                return out
            nxt = cur + step
            cur = nxt if (0 <= nxt < n and is_noise[nxt]) else start

    def predict(i, count):
        if i is None:
            raise ValueError("no noise window available to fill a gap")
        p = prev_valid(i)
        return circular(next_valid(i), count, 1) if p is None else circular(p, count, -1)

    order, empty, last = [], 0, None
    for i in range(n):
        if not is_noise[i]:
            empty += 1
            continue
        last = i
        if empty:
            order += predict(i, empty)
            empty = 0
        order.append(i)
    if empty:
        order += predict(last, empty)
    return order


def noise_profiler_predict(x, sample_rate=44100, time_window=0.1, percentile_level=95):
    """NoiseProfiler(x).getNoiseDataPredicted() as an array (denoise.py:104-106)."""
    x = np.asarray(x, dtype=np.float64)
    win = int(time_window * sample_rate)
    rms = window_rms(x, win)
    thr, is_noise = noise_windows(rms, percentile_level)
    order = predicted_noise_order(is_noise)
    return np.concatenate([x[i * win:(i + 1) * win] for i in order], axis=0), thr, is_noise


# ---------------------------------------------------------------------------------------------
# PCM file samples (the stage files between the phases are 16-bit PCM: denoise.py:75,90 and the
# sf.write calls of src/*.py go through libsndfile with clipping enabled).  libsndfile is not
# available to this build, so this restates its conversion and is NOT pinned by a reference run.
def pcm_encode(x, bits=16):
    """One sample at a time, integer arithmetic after the rounding: scale by 2^31, round half to
    even, saturate to int32, keep the top `bits` bits.  NaN -> 0."""
    out = np.empty(np.size(x), dtype=np.int64)
    for i, v in enumerate(np.asarray(x, dtype=np.float64).ravel()):
        if v != v:
            out[i] = 0
            continue
        s = v * 2147483648.0
        if s >= 2147483647.0:
            n = 2147483647
        elif s <= -2147483648.0:
            n = -2147483648
        else:
            n = int(round(s))               # Python's round() is round-half-to-even
        out[i] = n >> (32 - bits)           # arithmetic shift
    return out.reshape(np.shape(x))


def pcm_decode(q, bits=16):
    return np.asarray(q, dtype=np.float64) / float(1 << (bits - 1))
