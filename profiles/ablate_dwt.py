"""Timing-only ablation of the fused DWT tile kernels (ADN_OPT_ABLATE bits: 1 skip analysis,
2 skip shrink, 4 skip synthesis above level 1, 8 replace the level-1 synthesis arithmetic by
copies).  Outputs of ablated runs are wrong by construction; only times matter."""
import json, os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import get_engine
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
e = get_engine(0)
SR, n = 48000, 30 * 48000
x = torch.randn(clips * n, dtype=torch.float32, device="cuda")
bs = e.denoise_block_frames(SR, n)
starts = np.arange(0, n, bs, dtype=np.int64)
off = np.concatenate([c * n + starts for c in range(clips)])
lens = np.full(off.shape, bs, dtype=np.int64)
d_off, d_len = e._i64(off), e._i64(lens)
y = torch.empty_like(x)
for wavelet, level in (("db4", 2), ("db8", 4), ("db8", 8)):
    for abl in (0, 1, 2, 3, 4, 8, 12, 15):
        e.lib.adn_set_option(e._h, 2, abl)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=1, wavelet=wavelet, level=level, max_frames=bs)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(wavelet, level, "ablate", abl, "ms", round(min(ts), 3), flush=True)
e.lib.adn_set_option(e._h, 2, 0)
