"""Host time of one mono wavelet call (enqueue only, no synchronisation) against its device time."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
SR, clips, n = 48000, 256, 30 * 48000
x = torch.randn(clips * n, dtype=torch.float32, device="cuda") * 0.1
bs = e.denoise_block_frames(SR, n)
off = np.concatenate([c * n + np.arange(0, n, bs, dtype=np.int64) for c in range(clips)])
d_off, d_len = e._i64(off), e._i64(np.full(off.shape, bs, dtype=np.int64))
y = torch.empty_like(x)
for wavelet, level in (("db4", 2), ("db8", 8)):
    run = lambda: e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=1, wavelet=wavelet, level=level, max_frames=bs)
    run(); torch.cuda.synchronize()
    host = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); run(); host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20
    print(wavelet, level, "host enqueue ms", [round(h * 1e3, 3) for h in host], "wall per call ms %.3f" % (wall * 1e3), flush=True)
