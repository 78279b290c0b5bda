"""Engine: typed Python front end of the C ABI, torch tensors as device buffers."""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import AdnError, WAVELET, EXT, THR, AXIS

_ENGINES = {}


def get_engine(device=0):
    """Process-wide engine per device index."""
    if device not in _ENGINES:
        _ENGINES[device] = Engine(device)
    return _ENGINES[device]


def _suffix(dtype):
    if dtype == torch.float32:
        return "_f32"
    if dtype == torch.float64:
        return "_f64"
    raise TypeError("only float32 / float64 buffers are supported, got %s" % dtype)


_EMPTY = {}          # device -> small buffer whose address stands for empty tensors


def _ptr(t):
    """Device address of a tensor (NULL for None).  torch reports address 0 for an empty tensor; the
    C ABI reserves NULL for "argument not given", so an empty tensor passes a valid dummy address."""
    if t is None:
        return C.c_void_p(0)
    if t.numel() == 0:
        buf = _EMPTY.get(t.device)
        if buf is None:
            buf = _EMPTY[t.device] = torch.empty(64, dtype=torch.uint8, device=t.device)
        return C.c_void_p(buf.data_ptr())
    return C.c_void_p(t.data_ptr())


class Engine:
    """One adn_ctx bound to cuda:<device>.  Not thread-safe (the context owns a workspace)."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the MI355X HIP path has no CPU fallback")
        self.lib = _capi.load()
        self.device = int(device)
        self.torch_device = torch.device("cuda", self.device)
        h = C.c_void_p()
        self._check(self.lib.adn_create(C.byref(h), self.device))
        self._h = h
        self._desc_cache = {}
        self.use_current_stream()

    # ------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            raise AdnError(rc, self.lib.adn_last_error().decode("utf-8", "replace"))

    def use_current_stream(self):
        """Enqueue on torch's current stream of this device."""
        s = torch.cuda.current_stream(self.torch_device).cuda_stream
        self._check(self.lib.adn_set_stream(self._h, C.c_void_p(s)))

    def force_generic(self, on=True):
        """Test hook: run the shape-generic kernels instead of the specialised ones."""
        self._check(self.lib.adn_set_option(self._h, 1, int(bool(on))))

    def force_exact_fallback(self, on=True):
        """Test hook: the single-pass threshold statistics send every block to their exact
        per-block fallback selection (independent of force_generic)."""
        self._check(self.lib.adn_set_option(self._h, 3, int(bool(on))))

    def kernel_timing(self, on=True, only=None):
        """Record HIP events on the library's stream around every kernel launch, or around the
        launches of the one site called `only` (a name as kernel_times() reports it)."""
        self._check(self.lib.adn_kernel_timing_only(self._h, only.encode() if only else None))
        self._check(self.lib.adn_set_option(self._h, 4, int(bool(on))))

    def kernel_times(self):
        """{launch site: {"count": launches, "total_ms": summed duration}} since the last report
        (synchronises the stream)."""
        buf = C.create_string_buffer(1 << 16)
        n = int(self.lib.adn_kernel_times(self._h, buf, len(buf)))
        if n < -1:                                   # too small: the library kept the records, -n bytes hold them
            buf = C.create_string_buffer(-n)
            n = int(self.lib.adn_kernel_times(self._h, buf, len(buf)))
        if n < 0:
            raise AdnError(n, "adn_kernel_times failed")
        out = {}
        for line in buf.value.decode().splitlines():
            name, count, total = line.rsplit("\t", 2)
            out[name] = {"count": int(count), "total_ms": float(total)}
        return out

    def synchronize(self):
        self._check(self.lib.adn_synchronize(self._h))

    def workspace_bytes(self):
        return int(self.lib.adn_workspace_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.adn_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _i64(self, values):
        """int64 descriptor array on the device.  Descriptors repeat from call to call (same
        batch layout), so uploads are cached: steady-state calls do no host-to-device copy."""
        arr = np.ascontiguousarray(values, dtype=np.int64)
        key = arr.tobytes()
        t = self._desc_cache.get(key)
        if t is None:
            if len(self._desc_cache) >= 256:
                self._desc_cache.clear()
            t = torch.as_tensor(arr, device=self.torch_device)
            self._desc_cache[key] = t
        return t

    def _call(self, base, dtype, *args):
        # work goes to whatever stream torch has current now (callers may be inside a
        # torch.cuda.stream(...) block): the buffers torch hands out are ordered on that stream
        self.use_current_stream()
        fn = getattr(self.lib, base + _suffix(dtype))
        self._check(fn(self._h, *args))

    # ------------------------------------------------------- length helpers
    def wavelet_out_frames(self, frames):
        return int(self.lib.adn_wavelet_out_frames(int(frames)))

    def denoise_block_frames(self, rate, frames):
        return int(self.lib.adn_denoise_block_frames(int(rate), int(frames)))

    def stft_num_frames(self, n, n_fft, hop, padded=True):
        return int(self.lib.adn_stft_num_frames(int(n), int(n_fft), int(hop), int(bool(padded))))

    def istft_out_samples(self, n, n_fft, hop):
        return int(self.lib.adn_istft_out_samples(int(n), int(n_fft), int(hop)))

    # --------------------------------------------------------------- phase 2
    def wavelet_denoise(self, x, in_off, frames, out, out_off, n_ch=1, wavelet="db4", level=2,
                        thr_mode="soft", want_stats=False, max_frames=None, mode="per", axis="reference"):
        """Batched VisuShrink block denoise.  x / out: flat device tensors; in_off, frames,
        out_off: int64 device tensors (element offsets / frame counts per block).  axis="time"
        denoises every channel of a multi-channel block along time (statistics then hold one
        value per block and channel); the default does what the reference does."""
        nb = int(frames.numel())
        if max_frames is None:
            max_frames = int(frames.max().item()) if nb else 1
        sigma = thresh = zeroed = None
        if want_stats:
            nu = nb * n_ch if (n_ch > 1 and axis == "time") else nb
            sigma = torch.empty(nu, dtype=torch.float64, device=self.torch_device)
            thresh = torch.empty(nu, dtype=torch.float64, device=self.torch_device)
            zeroed = torch.empty(nu, dtype=torch.int64, device=self.torch_device)
        self._call("adn_wavelet_denoise", x.dtype, _ptr(x), _ptr(out), nb, _ptr(in_off), _ptr(frames),
                   _ptr(out_off), int(max_frames), int(n_ch), AXIS[axis], WAVELET[wavelet], int(level),
                   EXT[mode], THR[thr_mode], _ptr(sigma), _ptr(thresh), _ptr(zeroed))
        return (sigma, thresh, zeroed) if want_stats else None

    def denoise_signal(self, x, rate, wavelet="db4", level=2, thr_mode="soft", want_stats=False, mode="per",
                       axis="reference"):
        """AudioDeNoise.deNoise arithmetic on a device tensor of shape (frames,) or
        (frames, channels): blocks of int(rate * duration * 0.10) frames, each denoised
        independently.  Returns the concatenated blocks (and per-block stats)."""
        frames_total = int(x.shape[0])
        n_ch = 1 if x.dim() == 1 else int(x.shape[1])
        bs = self.denoise_block_frames(rate, frames_total)
        if bs <= 0:
            raise ValueError("input shorter than 10 frames: block size is zero")
        starts = np.arange(0, frames_total, bs, dtype=np.int64)
        lens = np.minimum(bs, frames_total - starts)
        if n_ch == 1 or axis == "time":
            out_lens = 2 * ((lens + 1) // 2)
        else:
            out_lens = lens
        out_starts = np.concatenate([[0], np.cumsum(out_lens)[:-1]])
        out = torch.empty((int(out_lens.sum()),) if n_ch == 1 else (int(out_lens.sum()), n_ch),
                          dtype=x.dtype, device=x.device)
        xc = x.contiguous()
        stats = self.wavelet_denoise(xc.view(-1), self._i64(starts * n_ch), self._i64(lens), out.view(-1),
                                     self._i64(out_starts * n_ch), n_ch=n_ch, wavelet=wavelet, level=level,
                                     thr_mode=thr_mode, want_stats=want_stats, max_frames=int(lens.max()),
                                     mode=mode, axis=axis)
        return (out, stats) if want_stats else out

    def wavedec(self, x, wavelet="db4", level=2, mode="per"):
        """pywt.wavedec of one 1-D device tensor -> list [cA_L, cD_L, ..., cD_1]."""
        n = int(x.numel())
        F = 8 if wavelet == "db4" else 16
        cap = n + (level + 1) * F + 2 * level + 8 if EXT[mode] else n + 2 * level + 8
        co = torch.empty(cap, dtype=x.dtype, device=x.device)
        lens = (C.c_int64 * (level + 1))()
        self._call("adn_wavedec", x.dtype, _ptr(x.contiguous()), n, WAVELET[wavelet], int(level), EXT[mode],
                   _ptr(co), cap, lens)
        out, pos = [], 0
        for ln in lens:
            out.append(co[pos:pos + ln].clone())
            pos += ln
        return out

    def waverec(self, coeffs, wavelet="db4", mode="per"):
        level = len(coeffs) - 1
        flat = torch.cat([c.reshape(-1) for c in coeffs]).contiguous()
        lens = (C.c_int64 * (level + 1))(*[int(c.numel()) for c in coeffs])
        cap = 2 * max(int(c.numel()) for c in coeffs) + 4
        y = torch.empty(cap, dtype=flat.dtype, device=flat.device)
        n_out = C.c_int64(0)
        self._call("adn_waverec", flat.dtype, _ptr(flat), lens, WAVELET[wavelet], level, EXT[mode], _ptr(y),
                   cap, C.byref(n_out))
        return y[:n_out.value].clone()

    # ------------------------------------------------------------ STFT stages
    @staticmethod
    def pack_signals(signals, n_ch=1, device=None):
        """Put files of any lengths into one flat buffer for a batched call, every file starting at a 16-byte
        aligned address (the fused kernels' 16-byte loads need that; files packed back to back at odd offsets take
        an element-wise path that costs 1.7 times the time).  signals: tensors of the same dtype, (frames,) or
        (frames, n_ch) each.  Returns (flat, lens, offsets): pass `lens` (frames per file) and `offsets=offsets`
        (element offsets) to the stage methods.  The gaps between files are zero."""
        dtype = signals[0].dtype
        device = signals[0].device if device is None else device
        per = 16 // signals[0].element_size()
        lens = np.array([int(s.shape[0]) for s in signals], dtype=np.int64)
        sizes = (lens * n_ch + per - 1) // per * per
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        flat = torch.zeros(int(sizes.sum()), dtype=dtype, device=device)
        for s, off in zip(signals, offsets):
            if s.dtype != dtype or s.numel() != s.shape[0] * n_ch:
                raise ValueError("signals must share a dtype and hold n_ch values per frame")
            flat[int(off):int(off) + s.numel()] = s.reshape(-1).to(device)
        return flat, lens, offsets

    def _signals(self, lens, n_ch, n_fft):
        lens = np.asarray(lens, dtype=np.int64)
        if (lens < n_fft).any():
            # scipy.signal.stft raises this for every such signal
            raise ValueError("window is longer than input signal")
        starts = np.concatenate([[0], np.cumsum(lens * n_ch)[:-1]]).astype(np.int64)
        return starts, lens

    def noise_psd(self, x, lens, n_ch=1, n_fft=2048, hop=512, padded=True, percentile=95.0,
                  spectral_floor=0.1, freq_bins=5, time_frames=5, offsets=None):
        """Phase 1 on a batch: x flat device tensor holding the signals back to back
        (or at `offsets`).  Returns (n_sig, n_fft/2+1) baseline magnitudes."""
        starts, lens = self._signals(lens, n_ch, n_fft)
        if offsets is not None:
            starts = np.asarray(offsets, dtype=np.int64)
        out = torch.empty((len(lens), n_fft // 2 + 1), dtype=x.dtype, device=x.device)
        d_off, d_len = self._i64(starts), self._i64(lens)     # keep alive across the call
        self._call("adn_noise_psd", x.dtype, _ptr(x), len(lens), _ptr(d_off), _ptr(d_len),
                   int(lens.max()), int(n_ch), int(n_fft), int(hop), int(bool(padded)), float(percentile),
                   float(spectral_floor), int(freq_bins), int(time_frames), _ptr(out))
        return out

    def band_summary(self, x, lens, sample_rate, bands, percentiles=(95.0, 50.0), n_ch=1, n_fft=2048, hop=512,
                     padded=True, offsets=None, want_time_mean=True, want_mag=False):
        """Phase-1 analysis summaries of a batch (src/analyze_audio.py:50-62, :161).  bands: list of
        (lo_hz, hi_hz).  Returns (mean_mag[n_sig, n_bands], perc_db[n_perc, n_sig, n_bands],
        time_mean[n_sig, T_max] or None[, mag[n_sig, F, T_max]]), the first two as float64 device tensors."""
        import ctypes as C
        starts, lens = self._signals(lens, n_ch, n_fft)
        if offsets is not None:
            starts = np.asarray(offsets, dtype=np.int64)
        nb, npc, ns = len(bands), len(percentiles), len(lens)
        lo = (C.c_double * max(nb, 1))(*[float(b[0]) for b in bands])
        hi = (C.c_double * max(nb, 1))(*[float(b[1]) for b in bands])
        pc = (C.c_double * max(npc, 1))(*[float(p) for p in percentiles])
        mean = torch.empty((ns, nb), dtype=torch.float64, device=x.device)
        perc = torch.empty((npc, ns, nb), dtype=torch.float64, device=x.device)
        t_max = self.stft_num_frames(int(lens.max()), n_fft, hop, padded)
        tmean = torch.zeros((ns, t_max), dtype=x.dtype, device=x.device) if want_time_mean else None
        d_off, d_len = self._i64(starts), self._i64(lens)
        mag = torch.zeros((ns, n_fft // 2 + 1, t_max), dtype=x.dtype, device=x.device) if want_mag else None
        self._call("adn_band_summary", x.dtype, _ptr(x), ns, _ptr(d_off), _ptr(d_len), int(lens.max()), int(n_ch),
                   int(sample_rate), int(n_fft), int(hop), int(bool(padded)), nb, lo, hi, npc, pc, _ptr(mean),
                   _ptr(perc), _ptr(tmean), _ptr(mag))
        return (mean, perc, tmean, mag) if want_mag else (mean, perc, tmean)

    def hpss(self, x, lens, sample_rate, n_ch=1, n_fft=2048, hop=512, kernel_size=31, margin_h=1.0, margin_p=5.0,
             power=2.0, mask_type="soft", bands=None, preserve_transients=True, threshold=0.3, blend=0.7,
             offsets=None):
        """Phase 3 (src/hpss_separate.py) on a batch.  Returns (y_harmonic, y_percussive, out_lens, stats) with
        the outputs back to back; stats[n_sig, 5] = harmonic RMS, percussive RMS, input RMS, max|y| before
        scaling, transient frames (float64 device tensor)."""
        import ctypes as C
        from ._capi import MASK
        starts, lens = self._signals(lens, n_ch, n_fft)
        if offsets is not None:
            starts = np.asarray(offsets, dtype=np.int64)
        out_lens = np.array([self.istft_out_samples(n, n_fft, hop) for n in lens], dtype=np.int64)
        out_off = np.concatenate([[0], np.cumsum(out_lens)[:-1]]).astype(np.int64)
        yh = self._out_buffer(int(out_lens.sum()), x)
        yp = self._out_buffer(int(out_lens.sum()), x)
        stats = torch.empty((len(lens), 5), dtype=torch.float64, device=x.device)
        bands = list(bands) if bands else []
        lo = (C.c_double * max(len(bands), 1))(*[float(b[0]) for b in bands])
        hi = (C.c_double * max(len(bands), 1))(*[float(b[1]) for b in bands])
        d_off, d_len, d_yoff = self._i64(starts), self._i64(lens), self._i64(out_off)
        self._call("adn_hpss", x.dtype, _ptr(x), len(lens), _ptr(d_off), _ptr(d_len), int(lens.max()), int(n_ch),
                   int(sample_rate), int(n_fft), int(hop), int(kernel_size), float(margin_h), float(margin_p),
                   float(power), MASK[mask_type], len(bands), lo, hi, int(bool(preserve_transients)),
                   float(threshold), float(blend), _ptr(yh), _ptr(yp), _ptr(d_yoff), _ptr(stats))
        return yh, yp, out_lens, stats

    def spectral_subtract(self, x, lens, noise_mag, n_ch=1, n_fft=2048, hop=512, alpha=2.2, beta=0.1,
                          smooth_time=3, smooth_freq=5, offsets=None, out=None, out_offsets=None, pcm16=None):
        """Phase 2b on a batch.  noise_mag: (F,) shared or (n_sig, F).  Returns (y, out_lens)
        with the outputs back to back in y (or at `out_offsets` of `out`).  `pcm16`: optional int16
        tensor laid out like y; it receives the 16-bit PCM file samples and y then holds the
        values read back from such a file."""
        starts, lens = self._signals(lens, n_ch, n_fft)
        if offsets is not None:
            starts = np.asarray(offsets, dtype=np.int64)
        out_lens = np.array([self.istft_out_samples(n, n_fft, hop) for n in lens], dtype=np.int64)
        if out_offsets is None:
            out_offsets = np.concatenate([[0], np.cumsum(out_lens)[:-1]]).astype(np.int64)
        if out is None:
            out = self._out_buffer(int(out_lens.sum()), x)
        noise_mag = noise_mag.to(dtype=x.dtype, device=x.device).contiguous()
        stride = 0 if noise_mag.dim() == 1 else noise_mag.shape[1]
        if noise_mag.shape[-1] != n_fft // 2 + 1:
            raise ValueError("noise profile must hold n_fft/2+1 bins (resize it on the host first)")
        d_off, d_len, d_yoff = self._i64(starts), self._i64(lens), self._i64(out_offsets)
        self._call("adn_spectral_subtract", x.dtype, _ptr(x), len(lens), _ptr(d_off),
                   _ptr(d_len), int(lens.max()), int(n_ch), _ptr(noise_mag), int(stride), int(n_fft),
                   int(hop), float(alpha), float(beta), int(smooth_time), int(smooth_freq), _ptr(out),
                   _ptr(d_yoff), _ptr(self._pcm_arg(pcm16, out)))
        return out, out_lens

    def wiener(self, x, lens, sample_rate, n_ch=1, n_fft=2048, hop=512, update_ms=20, kernel_bins=5,
               vad_guided=True, offsets=None, out=None, out_offsets=None, want_gain=False, pcm16=None):
        """Phase 2c on a batch.  Returns (y, out_lens[, gain]).  `pcm16` as for spectral_subtract."""
        starts, lens = self._signals(lens, n_ch, n_fft)
        if offsets is not None:
            starts = np.asarray(offsets, dtype=np.int64)
        out_lens = np.array([self.istft_out_samples(n, n_fft, hop) for n in lens], dtype=np.int64)
        if out_offsets is None:
            out_offsets = np.concatenate([[0], np.cumsum(out_lens)[:-1]]).astype(np.int64)
        if out is None:
            out = self._out_buffer(int(out_lens.sum()), x)
        gain = torch.empty((len(lens), n_fft // 2 + 1), dtype=x.dtype, device=x.device) if want_gain else None
        d_off, d_len, d_yoff = self._i64(starts), self._i64(lens), self._i64(out_offsets)
        self._call("adn_wiener", x.dtype, _ptr(x), len(lens), _ptr(d_off), _ptr(d_len),
                   int(lens.max()), int(n_ch), int(sample_rate), int(n_fft), int(hop), float(update_ms),
                   int(kernel_bins), int(bool(vad_guided)), _ptr(out), _ptr(d_yoff), _ptr(gain),
                   _ptr(self._pcm_arg(pcm16, out)))
        return (out, out_lens, gain) if want_gain else (out, out_lens)

    @staticmethod
    def _out_buffer(total, like):
        """Flat output tensor of `total` samples (a signal exactly one window long yields no output
        samples at all: scipy's istft trim)."""
        return torch.empty(total, dtype=like.dtype, device=like.device)

    @staticmethod
    def _pcm_arg(pcm16, out):
        if pcm16 is None:
            return None
        if pcm16.dtype != torch.int16 or pcm16.numel() < out.numel() or not pcm16.is_contiguous() \
                or pcm16.device != out.device:
            raise ValueError("pcm16 must be a contiguous int16 device tensor at least as long as the output")
        return pcm16

    def mad(self, x, lens=None):
        """median(|x - median(x)|) per segment of a flat device tensor (one segment when
        `lens` is None).  Returns a float64 device tensor of one value per segment."""
        lens = np.asarray([x.numel()] if lens is None else lens, dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        out = torch.empty(len(lens), dtype=torch.float64, device=x.device)
        d_off, d_len = self._i64(offs), self._i64(lens)
        self._call("adn_mad", x.dtype, _ptr(x.contiguous()), len(lens), _ptr(d_off), _ptr(d_len),
                   int(lens.max()), _ptr(None), _ptr(out))
        return out if len(lens) > 1 else out[0]

    # ------------------------------------------ pieces for recordings split into time chunks
    def wavelet_shrink(self, x, in_off, frames, out, out_off, thresh, wavelet="db4", level=2, thr_mode="soft",
                       mode="per", max_frames=None, want_zeroed=False):
        """Mono blocks shrunk with the thresholds in `thresh` (float64 device tensor, one per block)."""
        nb = int(frames.numel())
        if max_frames is None:
            max_frames = int(frames.max().item()) if nb else 1
        zeroed = torch.empty(nb, dtype=torch.int64, device=self.torch_device) if want_zeroed else None
        self._call("adn_wavelet_shrink", x.dtype, _ptr(x), _ptr(out), nb, _ptr(in_off), _ptr(frames), _ptr(out_off),
                   int(max_frames), WAVELET[wavelet], int(level), EXT[mode], THR[thr_mode], _ptr(thresh), _ptr(zeroed))
        return zeroed

    def select_hist(self, vals, offs, lens, prefix, done_bits, hist, absdev=False, centers=None, max_len=None):
        """One round of the distributed order statistic: `hist` (int32 [n_seg, 256]) receives the counts of the next
        key digit of this holder's values under `prefix` (uint64 keys as int64 tensor)."""
        n_seg = int(lens.numel())
        if max_len is None:
            max_len = int(lens.max().item()) if n_seg else 0
        self._call("adn_select_hist", vals.dtype, _ptr(vals), n_seg, _ptr(offs), _ptr(lens), int(max_len),
                   int(bool(absdev)), _ptr(centers), _ptr(prefix), int(done_bits), _ptr(hist))

    def select_resolve(self, hist, key_bits, done_bits, rank, prefix):
        self.use_current_stream()
        self._check(self.lib.adn_select_resolve(self._h, _ptr(hist), int(rank.numel()), int(key_bits), int(done_bits),
                                                _ptr(rank), _ptr(prefix)))

    def select_value(self, prefix, dtype):
        out = torch.empty(prefix.numel(), dtype=torch.float64, device=self.torch_device)
        self._call("adn_select_value", dtype, _ptr(prefix), int(prefix.numel()), _ptr(out))
        return out

    def select_combine(self, lo, hi, n_values, q, dtype):
        """Median (q=None) or percentile q of segments of n_values values from the two bracketing order statistics."""
        out = torch.empty_like(lo)
        self._call("adn_select_combine", dtype, _ptr(lo), _ptr(hi), int(lo.numel()), int(n_values),
                   -1.0 if q is None else float(q), _ptr(out))
        return out

    def stft_rows(self, x, lens, n_ch=1, n_fft=2048, hop=512, padded=True, power=True, offsets=None):
        """|Z|^2 (or |Z|) rows of a batch: returns (S[n_sig, F, Ts], frame counts)."""
        starts, lens = self._signals(lens, n_ch, n_fft)
        if offsets is not None:
            starts = np.asarray(offsets, dtype=np.int64)
        frames = np.array([self.stft_num_frames(n, n_fft, hop, padded) for n in lens], dtype=np.int64)
        ts = (int(frames.max()) + 31) // 32 * 32
        S = torch.zeros((len(lens), n_fft // 2 + 1, ts), dtype=x.dtype, device=x.device)
        d_off, d_len = self._i64(starts), self._i64(lens)
        self._call("adn_stft_rows", x.dtype, _ptr(x), len(lens), _ptr(d_off), _ptr(d_len), int(lens.max()), int(n_ch),
                   int(n_fft), int(hop), int(bool(padded)), int(bool(power)), _ptr(S), ts)
        return S, frames

    def vad_frames(self, buf, base_sample, n_total, sample_rate, t0, n_frames, update_ms=20, hop=512, n_ch=1):
        """Energy-VAD flags of STFT frames t0 .. t0 + n_frames - 1 of a signal of n_total samples; `buf` holds the
        signal's samples from `base_sample` on."""
        vad = torch.empty(int(n_frames), dtype=torch.uint8, device=buf.device)
        n_buf = int(buf.shape[0]) if buf.dim() > 1 else int(buf.numel()) // n_ch
        self._call("adn_vad_frames", buf.dtype, _ptr(buf), n_buf, int(n_ch), int(base_sample), int(n_total),
                   int(sample_rate), float(update_ms), int(hop), int(t0), int(n_frames), _ptr(vad))
        return vad

    def rows_masked_sum(self, S, row_len, mask):
        """(sums[n_rows] float64, count) over the masked frames of the rows of S[n_rows, stride]."""
        n_rows, stride = int(S.shape[0]), int(S.shape[1])
        sums = torch.empty(n_rows, dtype=torch.float64, device=S.device)
        count = torch.zeros(1, dtype=torch.int64, device=S.device)
        self._call("adn_rows_masked_sum", S.dtype, _ptr(S), n_rows, int(row_len), stride, _ptr(mask), _ptr(sums),
                   _ptr(count))
        return sums, count

    def wiener_gain(self, noise_psd, speech_psd, dtype, kernel_bins=5):
        """Per-bin gain rows from float64 [n_sig, F] statistics."""
        n_sig, F = int(noise_psd.shape[0]), int(noise_psd.shape[1])
        gain = torch.empty((n_sig, F), dtype=dtype, device=noise_psd.device)
        self._call("adn_wiener_gain", dtype, _ptr(noise_psd.contiguous()), _ptr(speech_psd.contiguous()), n_sig, F,
                   int(kernel_bins), _ptr(gain))
        return gain

    def stft_apply_gain(self, x, lens, gain, n_ch=1, n_fft=2048, hop=512, offsets=None):
        """STFT -> Z * gain -> ISTFT -> clip.  gain: (F,) shared or (n_sig, F).  Returns (y, out_lens)."""
        starts, lens = self._signals(lens, n_ch, n_fft)
        if offsets is not None:
            starts = np.asarray(offsets, dtype=np.int64)
        out_lens = np.array([self.istft_out_samples(n, n_fft, hop) for n in lens], dtype=np.int64)
        out_offsets = np.concatenate([[0], np.cumsum(out_lens)[:-1]]).astype(np.int64)
        out = self._out_buffer(int(out_lens.sum()), x)
        gain = gain.to(dtype=x.dtype, device=x.device).contiguous()
        stride = 0 if gain.dim() == 1 else gain.shape[1]
        d_off, d_len, d_yoff = self._i64(starts), self._i64(lens), self._i64(out_offsets)
        self._call("adn_stft_apply_gain", x.dtype, _ptr(x), len(lens), _ptr(d_off), _ptr(d_len), int(lens.max()),
                   int(n_ch), int(n_fft), int(hop), _ptr(gain), int(stride), _ptr(out), _ptr(d_yoff), _ptr(None))
        return out, out_lens

    # ------------------------------------------------------------------ misc
    def window_rms(self, x, win):
        n_rows = int(x.shape[0])
        n_ch = 1 if x.dim() == 1 else int(x.shape[1])
        n_win = -(-n_rows // int(win))
        out = torch.empty(n_win, dtype=torch.float64, device=x.device)
        self._call("adn_window_rms", x.dtype, _ptr(x.contiguous()), n_rows, n_ch, int(win), _ptr(out))
        return out

    def pcm_encode(self, x, bits=16, dequantize=False):
        """Integer PCM sample values of `x` (int16 for 16 bits, int32 for 24 / 32) as the stage
        files hold them; with `dequantize` also the values a reader of the file gets back
        (True: a new tensor, "inplace": written over `x`).  Returns q or (q, xq)."""
        x = x.contiguous()
        q = torch.empty(x.shape, dtype=torch.int16 if bits == 16 else torch.int32, device=x.device)
        xq = x if dequantize == "inplace" else (torch.empty_like(x) if dequantize else None)
        self._call("adn_pcm_encode", x.dtype, _ptr(x), x.numel(), int(bits), _ptr(q), _ptr(xq))
        return (q, xq) if dequantize else q

    def synth_noisy_speech(self, n, sample_rate, seed=42, stream=0, sigma=0.05, speech_gain=1.0,
                           start=0, dtype=torch.float64, out=None):
        if out is None:
            out = torch.empty(int(n), dtype=dtype, device=self.torch_device)
        self._call("adn_synth_noisy_speech", out.dtype, _ptr(out), int(n), int(start), int(sample_rate),
                   C.c_uint64(seed), C.c_uint64(stream), float(sigma), float(speech_gain))
        return out
