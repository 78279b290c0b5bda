"""Diagnostic: time the shape-generic (binary64) fused STFT kernel with individual stages skipped
(ADN_OPT_ABLATE).  Outputs of ablated runs are wrong by construction; only times matter.
bits: 1 analysis, 32 magnitudes only, 128 forward FFT only, 256 real split only, 2 synthesis,
64 modify only, 4 overlap-add."""
import os, sys, time, json
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
B, n, SR = 16, 60 * 48000, 48000
x = torch.randn(B * n, device="cuda", dtype=torch.float64) * 0.1
noise = torch.rand(1025, device="cuda", dtype=torch.float64) * 0.01
for mode in ("subtract", "wiener"):
    for abl in (0, 1, 32, 32 + 256, 32 + 256 + 128, 2, 64, 4, 7):
        e.lib.adn_set_option(e._h, 2, abl)
        ts = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if mode == "subtract":
                e.spectral_subtract(x, [n] * B, noise)
            else:
                e.wiener(x, [n] * B, SR)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(mode, "ablate", abl, "ms", round(min(ts), 3), flush=True)
e.lib.adn_set_option(e._h, 2, 0)
