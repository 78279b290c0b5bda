"""Diagnostics: the mono wavelet stage (256 x 30 s clips, db4 level 2) with parts of the tile kernel switched off
(option ADN_OPT_ABLATE bits 1: no analysis, 2: no shrink, 4: only the last synthesis level, 8: synthesis copies)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
clips, SR, n = 256, 48000, 30 * 48000
x = torch.randn(clips * n, dtype=torch.float32, device="cuda") * 0.1
bs = e.denoise_block_frames(SR, n)
off = np.concatenate([c * n + np.arange(0, n, bs, dtype=np.int64) for c in range(clips)])
d_off, d_len = e._i64(off), e._i64(np.full(off.shape, bs, dtype=np.int64))
y = torch.empty_like(x)
wavelet, level = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("db4", 2)
run = lambda: e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=1, wavelet=wavelet, level=level, max_frames=bs)
for ab in (0, 1, 2, 8, 9):
    e.lib.adn_set_option(e._h, 2, ab)
    run(); torch.cuda.synchronize()
    e.kernel_timing(True)
    for _ in range(3):
        run()
    kt = e.kernel_times()
    e.kernel_timing(False)
    print(wavelet, level, "ablate", ab, {k[:24]: round(v["total_ms"] / 3, 3) for k, v in kt.items() if v["total_ms"] / 3 > 0.05})
e.lib.adn_set_option(e._h, 2, 0)
