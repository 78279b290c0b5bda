"""Per-kernel times of the mono wavelet stage (BASELINE config 3 shape: 256 x 30 s 48 kHz mono clips, 10 blocks each)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd"))
from adn_hip import get_engine
wavelet, level = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("db4", 2)
dt = torch.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else torch.float32
e = get_engine(0)
if os.environ.get("ADN_ABLATE"):              # diagnostics: alternative routes of the mono statistics (512: two passes with the finest band in memory)
    e._check(e.lib.adn_set_option(e._h, 2, int(os.environ["ADN_ABLATE"])))
clips, SR, n = 256, 48000, 30 * 48000
x = torch.empty(clips * n, dtype=dt, device="cuda")
tmp = torch.empty(n, dtype=torch.float64, device="cuda")
for c in range(16):
    e.synth_noisy_speech(n, SR, stream=c, out=tmp)
    x[c * n:(c + 1) * n] = tmp.to(dt)
for c in range(16, clips):
    x[c * n:(c + 1) * n] = x[(c % 16) * n:(c % 16 + 1) * n]
bs = e.denoise_block_frames(SR, n)
off = np.concatenate([c * n + np.arange(0, n, bs, dtype=np.int64) for c in range(clips)])
d_off, d_len = e._i64(off), e._i64(np.full(off.shape, bs, dtype=np.int64))
y = torch.empty_like(x)
run = lambda: e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=1, wavelet=wavelet, level=level, max_frames=bs)
run(); torch.cuda.synchronize()
e.kernel_timing(True)
for _ in range(5):
    run()
kt = e.kernel_times()
e.kernel_timing(False)
tot = 0.0
for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"]):
    print("%-40s %8.4f ms/call x %g" % (k[:40], v["total_ms"] / v["count"], v["count"] / 5))
    tot += v["total_ms"] / 5
print("sum of kernels per call: %.4f ms" % tot)
