"""Secondary benchmark: BASELINE.json config 3 - batch of N x 30 s 48 kHz mono clips,
db8 level 8 VisuShrink (10 blocks per clip), one MI355X.  Prints one JSON line."""
import json, os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import get_engine
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wavelet, level = (sys.argv[2], int(sys.argv[3])) if len(sys.argv) > 3 else ("db8", 8)
steps = 5
e = get_engine(0)
SR, n = 48000, 30 * 48000
x = torch.empty(clips * n, dtype=torch.float32, device="cuda")
tmp = torch.empty(n, dtype=torch.float64, device="cuda")
for c in range(min(clips, 16)):
    e.synth_noisy_speech(n, SR, stream=c, out=tmp)
    x[c * n:(c + 1) * n] = tmp.float()
for c in range(16, clips):                       # reuse the 16 generated clips
    x[c * n:(c + 1) * n] = x[(c % 16) * n:(c % 16 + 1) * n]
bs = e.denoise_block_frames(SR, n)
starts = np.arange(0, n, bs, dtype=np.int64)
off = np.concatenate([c * n + starts for c in range(clips)])
lens = np.full(off.shape, bs, dtype=np.int64)
d_off, d_len = e._i64(off), e._i64(lens)
y = torch.empty_like(x)
def run():
    e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=1, wavelet=wavelet, level=level, max_frames=bs)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
alg = 2 * 4 * clips * n
print(json.dumps({"workload": "%d x 30 s 48 kHz mono, %s level %d VisuShrink" % (clips, wavelet, level),
                  "ms": dt * 1e3, "audio_sec_per_sec": clips * 30 / dt, "algorithmic_GBps": alg / dt / 1e9,
                  "hbm_frac": alg / dt / 1e9 / 8000, "workspace_GB": e.workspace_bytes() / 1e9}))
