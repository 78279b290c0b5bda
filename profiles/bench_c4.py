"""Secondary benchmark: BASELINE.json config 4 shape on ONE rank - a shard of N x 30 s 48 kHz mono files,
spectral subtraction + Wiener back to back with one shared noise profile (learnt once from a 5 s noise
sample, as the owner rank would before broadcasting it).  Files are independent, so N ranks each
run this on their own shard.  Prints one JSON line."""
import json, os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import get_engine
files = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = 5
e = get_engine(0)
SR, n = 48000, 30 * 48000
x = torch.empty(files * n, dtype=torch.float32, device="cuda")
tmp = torch.empty(n, dtype=torch.float64, device="cuda")
for c in range(min(files, 16)):
    e.synth_noisy_speech(n, SR, stream=c, out=tmp)
    x[c * n:(c + 1) * n] = tmp.float()
for c in range(16, files):
    x[c * n:(c + 1) * n] = x[(c % 16) * n:(c % 16 + 1) * n]
noise = e.synth_noisy_speech(5 * SR, SR, stream=100000, speech_gain=0.0).float()
base = e.noise_psd(noise, [5 * SR])[0]
lens = [n] * files
l2b = e.istft_out_samples(n, 2048, 512)
l2c = e.istft_out_samples(l2b, 2048, 512)
y2b = torch.empty(files * l2b, dtype=torch.float32, device="cuda")
y2c = torch.empty(files * l2c, dtype=torch.float32, device="cuda")
off_b = np.arange(files, dtype=np.int64) * l2b
off_c = np.arange(files, dtype=np.int64) * l2c
ev = {}
def run(timed=False):
    for name, fn in (("2b", lambda: e.spectral_subtract(x, lens, base, out=y2b, out_offsets=off_b)),
                     ("2c", lambda: e.wiener(y2b, [l2b] * files, SR, offsets=off_b, out=y2c, out_offsets=off_c))):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        if timed:
            ev.setdefault(name, []).append((a, b))
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    run(True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
ms = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in ev.items()}
alg = 2 * 4 * files * (n + l2b)
print(json.dumps({"workload": "%d x 30 s 48 kHz mono files, spectral subtraction + Wiener, shared noise profile" % files,
                  "ms": dt * 1e3, "stage_ms": ms, "audio_sec_per_sec": files * 30 / dt,
                  "algorithmic_GBps": alg / dt / 1e9, "hbm_frac": alg / dt / 1e9 / 8000}))
