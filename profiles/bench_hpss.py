"""Secondary benchmark: phase 3 HPSS (adn_hpss) on a batch of mono 60 s 48 kHz clips, fp32,
kernel 31, soft masks, transient preservation.  Prints one JSON line."""
import json, os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import get_engine
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
e = get_engine(0)
SR, n = 48000, 60 * 48000
x = torch.randn(clips * n, dtype=torch.float32, device="cuda") * 0.1
def run():
    return e.hpss(x, [n] * clips, SR)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
alg = 4 * clips * n * 3          # mono in, two mono outputs
print(json.dumps({"workload": "%d x 60 s 48 kHz mono, HPSS kernel 31, soft masks, transient preservation" % clips,
                  "ms": dt * 1e3, "audio_sec_per_sec": clips * 60 / dt, "algorithmic_GBps": alg / dt / 1e9,
                  "hbm_frac": alg / dt / 1e9 / 8000}))
