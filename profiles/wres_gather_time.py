"""Timing of the statistics kernels of mono blocks (256 x 30 s clips): wave-resident gather against the direct one."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
SR = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
clips, n = 256, 30 * SR
dt = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else torch.float32
x = torch.randn(clips * n, dtype=dt, device="cuda") * 0.1
bs = e.denoise_block_frames(SR, n)
off = np.concatenate([c * n + np.arange(0, n, bs, dtype=np.int64) for c in range(clips)])
d_off, d_len = e._i64(off), e._i64(np.full(off.shape, bs, dtype=np.int64))
y = torch.empty_like(x)
for wavelet, level in (("db4", 2), ("db8", 8)):
    for ab in (0, 4096):
        e.lib.adn_set_option(e._h, 2, ab)
        run = lambda: e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=1, wavelet=wavelet, level=level, max_frames=bs)
        run(); torch.cuda.synchronize()
        e.kernel_timing(True)
        for _ in range(3):
            run()
        kt = e.kernel_times()
        e.kernel_timing(False)
        tot = sum(v["total_ms"] for v in kt.values()) / 3
        print(wavelet, level, "ablate", ab, "total %.3f" % tot, {k[:26]: round(v["total_ms"] / 3, 3) for k, v in kt.items() if v["total_ms"] / 3 > 0.002}, flush=True)
e.lib.adn_set_option(e._h, 2, 0)
