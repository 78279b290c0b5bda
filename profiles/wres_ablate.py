"""Timing-only variants of the wave-resident wavelet tiles (ADN_OPT_ABLATE bits 8192: no output stores, 16384: stores
one sample early, i.e. 16-byte aligned)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
SR, clips, n = 48000, 256, 30 * 48000
x = torch.randn(clips * n + 64, dtype=torch.float32, device="cuda")[64:] * 0.1
bs = e.denoise_block_frames(SR, n)
off = np.concatenate([c * n + np.arange(0, n, bs, dtype=np.int64) for c in range(clips)])
d_off, d_len = e._i64(off), e._i64(np.full(off.shape, bs, dtype=np.int64))
y = torch.empty(clips * n + 64, dtype=torch.float32, device="cuda")[64:]
for wavelet, level in (("db4", 2), ("db8", 8)):
    for ab in (0, 8192, 16384):
        e.lib.adn_set_option(e._h, 2, ab)
        run = lambda: e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=1, wavelet=wavelet, level=level, max_frames=bs)
        run(); torch.cuda.synchronize()
        e.kernel_timing(True)
        for _ in range(3):
            run()
        kt = e.kernel_times()
        e.kernel_timing(False)
        print(wavelet, level, "ablate", ab, {k[:28]: (round(v["total_ms"] / v["count"], 3), v["count"] // 3) for k, v in kt.items() if "wres" in k or "k_tile" in k or "deep" in k}, flush=True)
e.lib.adn_set_option(e._h, 2, 0)
